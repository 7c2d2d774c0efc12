/*
 * disconet_hip.h -- C ABI of libdisconet_hip.so, the MI355X (gfx950) kernels of
 * the coperception `--com disco` detector hot path.
 *
 * The reference has NO FFI for this path: it is pure Python over torch ops
 * (SURVEY.md §2.2, §8(b)); its source is not in the mount
 * (/root/reference/coperception/ is an empty submodule dir,
 * /root/reference/.gitmodules:1-3), so each entry point cites the upstream
 * function it replaces by path (no line numbers exist to cite) and the mounted
 * call sites that reach it: /root/reference/README.md:54-63 (train_codet.py
 * --com disco) and README.md:68-75 (test_codet.py --com disco).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller owns every buffer (incl. workspaces); nothing is allocated,
 *     nothing synchronises; all work is enqueued on `stream` (a hipStream_t
 *     passed as void*, NULL = the default stream);
 *   - activations are channels-last.  Two storage forms: float32 NHWC [image][y][x][channel]
 *     (dn_conv2d*, the warp / fusion kernels' maps, the heads' outputs, everything in
 *     disconet_train.h) and the split-planar "SP" form of the inference conv engine
 *     (dn_spconv2d*, see "SP tensor" below: f16 hi/lo planes, opaque bytes);
 *     dn_sp_from_nhwc / dn_sp_to_nhwc convert.  Images of a batch are agent-major
 *     (image = agent * B + b), the order the reference's tools build with torch.cat over agents;
 *   - ONE stream per device: kernels of this library must not run CONCURRENTLY with each other
 *     (two streams, two graphs in flight).  Beside the split-f16 conv kernels another kernel has
 *     been observed to compute with corrupted VGPR lanes; calls enqueued on
 *     one stream, or ordered by events, are safe;
 *   - return 0 on success, negative on error; dn_last_error() returns a
 *     thread-local message for the last failing call on this thread;
 *   - re-entrant.  Process-wide state is limited to launch-time caches filled on first use
 *     (kernel attributes / occupancy per instantiation) and the tools-only knobs
 *     dn_spconv_force_config() and the DN_* environment variables read once.
 */
#ifndef DISCONET_HIP_H
#define DISCONET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DN_OK 0
#define DN_ERR_ARG (-1)
#define DN_ERR_LAUNCH (-2)
#define DN_ERR_UNSUPPORTED (-3)

int dn_version(void);
/* 16 hex digits: SHA-256 over every csrc source / header, include/*.h and the compiler flags the library was built
 * from (disconet_amd/csrc/build.py :: tree_hash).  The Python binding refuses a library whose id differs from the
 * tree it sits in -- a stale .so fails loudly instead of running kernels no commit reproduces. */
const char* dn_build_id(void);
const char* dn_last_error(void);

/* Range guard of the split-f16 engines (dn_spconv2d*, dn_sp_from_nhwc, dn_disco_fuse_mlp).  A value is
 * stored as hi + lo f16 halves, so the format has the f16 EXPONENT range:
 *   - a split clamps at +-65504 (the result then no longer follows the fp32 reference);
 *   - PRECISION FLOOR: `lo = half(x - hi)` is a subnormal once |x| < ~0.125, so below that magnitude an operand
 *     carries an ABSOLUTE error floor of 2^-25 (~3e-8) instead of 2^-22 relative (18 significant bits at
 *     |x| ~ 0.02, 14 at ~1e-3).  Weights are lifted out of that range at pack time by a power of two folded
 *     into the layer's scale (exact); activations are not lifted -- post-ReLU maps of O(1) meet the 1e-4 parity
 *     bar with three orders of margin, maps that are uniformly tiny (|x| << 1e-3) should be rescaled through
 *     their BatchNorm.
 * Every kernel that splits values keeps a sticky word in device memory:
 *   bit 1 (2): a value with |x| > 2^14 was split -- within two binades of the limit, rescale;
 *   bit 0 (1): a value was clamped to +-65504 -- results of this device since the last reset are wrong;
 *   bit 2 (4): a NaN was split by dn_sp_from_nhwc (input) or came out of dn_disco_fuse_mlp (inf / inf in the softmax).
 *              ReLU and the clamp turn a NaN into a finite number, so without this bit it would vanish.  The conv
 *              epilogues do NOT test (it cost ~1 % of the step): out of finite operands they cannot produce a NaN
 *              (overflow is clamped and flagged by bit 0) -- the caller must not pack non-finite weights / scale /
 *              shift (the Python host refuses them when it packs a plan).
 * dn_sp_range_flags: the OR over the library's kernels on the current device; with reset != 0 it clears them.
 *   BLOCKING (hipDeviceSynchronize + a device -> host copy): validation time.  Bit 31 (0x80000000) set = the READ
 *   failed (synchronisation, allocation or copy error): the flags are unknown -- treat as an error, never as "clean"
 *   (the Python host raises).
 * dn_sp_range_flags_async: the same OR enqueued on `stream` into *dst_device (a device word the caller zeroed, or
 *   `reset` bit 1 set: zeroed here first by a kernel launch -- what a CAPTURED step uses, a memset node does not replay
 *   reliably on ROCm 7.2); `reset` bit 0 clears the sticky flags.  No synchronisation, legal inside a stream capture.
 *   A flagged forward's OUTPUTS ARE INVALID (bit 0: clamped values; bit 2: a NaN was turned into a finite number): the
 *   caller must discard them, not only log the flag.  The Python host enqueues it behind a forward, copies the
 *   word to pinned memory and raises at the next call once the copy has landed: the guard is on by default and
 *   costs nothing but four tiny launches (DN_SP_CHECK=0 switches it off, =1 makes every forward block and check). */
unsigned dn_sp_range_flags(int reset);
int dn_sp_range_flags_async(unsigned* dst_device, int reset, void* stream);

/* ------------------------------------------------------------------------
 * K1 -- point cloud -> BEV occupancy.
 * Replaces upstream:coperception/utils/data_util.py :: voxelize_occupy
 * (Appx A.2): strict extent filter on raw floats,
 * floor(xyz / voxel_size) with a float64 divide, index = q - floor(lo/voxel).
 *   pts        [n_pts][pt_stride] float32, x,y,z in columns 0..2
 *   voxel_size_host[3], extents_host[6] = {xlo,xhi,ylo,yhi,zlo,zhi}  (host doubles)
 *   dims_host[3]   = grid dims (X, Y, Z); dense is [X][Y][Z] float32, set to
 *                    0/1 by this call (it zero-fills first).
 * ------------------------------------------------------------------------ */
int dn_voxelize_occupy(const float* pts, int n_pts, int pt_stride,
                       const double* voxel_size_host, const double* extents_host,
                       const int* dims_host, float* dense, void* stream);

/* Sorted-unique voxel index list of a dense grid, in the reference's
 * lexsort(x, then y, then z) order (= linear order of the [X][Y][Z] grid).
 *   indices   [capacity][3] int32 out;  count: int32 out (device)
 *   workspace: dn_voxel_compact_workspace(dims) bytes. */
size_t dn_voxel_compact_workspace(const int* dims_host);
int dn_voxel_compact(const float* dense, const int* dims_host, int32_t* indices,
                     int capacity, int32_t* count, void* workspace, void* stream);

/* Replaces upstream:coperception/datasets/V2XSimDet.py :: __getitem__ (dense
 * rebuild) for a whole batch: image g owns rows
 * offsets[g]..offsets[g+1] of indices[.][3]; dense[g][X][Y][Z] = 1 there, else 0. */
int dn_scatter_dense(const int32_t* indices, const int32_t* offsets, int n_images,
                     int n_indices_total, const int* dims_host, float* dense, void* stream);
/* The same rebuild written in the conv engine's split-planar layout (see "SP tensor" below):
 * dense_sp is the SP tensor [n_images][ceil(Z/16)][4][X][Y] x 16 bytes of the bevs batch. */
int dn_scatter_dense_sp(const int32_t* indices, const int32_t* offsets, int n_images,
                        int n_indices_total, const int* dims_host, void* dense_sp, void* stream);
/* ... and as a HI-ONLY SP tensor [n_images][ceil(Z/16)][2 octets][X][Y] x 16 bytes (half the bytes): an
 * occupancy grid is exact in binary16, its lo planes would be all zero.  dn_spconv2d reads it as
 * source 0 of a 3x3 stride-1 layer when dn_conv_desc.math == 3 (half the operand traffic, two MFMAs per
 * product instead of three; the results are bit-identical to the full form). */
int dn_scatter_dense_sp_hi(const int32_t* indices, const int32_t* offsets, int n_images,
                           int n_indices_total, const int* dims_host, void* dense_sp_hi, void* stream);
/* ... and as an occupancy BIT grid bits[n_images][X][Y] (uint32; bit z = height bin z holds a point; Z <= 32):
 * 1/32 of the float32 grid ("520 KB/scene as bits").  dn_spconv2d reads it as source 0 of a 3x3
 * stride-1 layer with c_out <= 32 when dn_conv_desc.math == 4: the words are expanded to the hi-only form's halves on
 * their way into LDS, the arithmetic and every result are those of math == 3 on the expanded grid. */
int dn_scatter_dense_bits(const int32_t* indices, const int32_t* offsets, int n_images,
                          int n_indices_total, const int* dims_host, uint32_t* bits, void* stream);

/* ------------------------------------------------------------------------
 * K2/K3/K7 -- implicit-GEMM convolution on fp32 MFMA with fused
 * (bias + eval-BatchNorm) affine and ReLU epilogue.
 * Replaces the conv2d/conv3d(1,1,1) + batch_norm + relu (+ interpolate x2
 * nearest + cat) sequences of upstream:coperception/models/det/backbone/
 * Backbone.py :: Backbone.encode / .decode and of ClassificationHead /
 * SingleRegressionHead (Appx A.6).
 *
 * The logical input is cat([up(src0), src1], channel): src0 has c0 channels
 * and, when up0 != 0, is stored at (h_in/2, w_in/2) and nearest-upsampled x2
 * on the fly; src1 (c1 channels, may be 0/NULL) is at (h_in, w_in).
 * ------------------------------------------------------------------------ */
typedef struct dn_conv_desc {
  int32_t n_images;
  int32_t h_in, w_in;    /* logical conv input size (after the x2 upsample) */
  int32_t c0, c1;        /* channels taken from src0 / src1 */
  int32_t up0;           /* src0 is half resolution, upsample x2 nearest */
  int32_t c_out;
  int32_t ksize;         /* 1 or 3 (padding = ksize/2) */
  int32_t stride;        /* 1 or 2 */
  int32_t relu;          /* apply ReLU after the affine */
  int32_t ld0, ld1, ldo; /* floats per pixel of src0 / src1 / out (>= channels) */
  int32_t math;          /* 0 = exact fp32 MFMA; 1 = split-f16 (x = hi + lo halves, hi*hi + hi*lo +
                            lo*hi on the f16 MFMA, fp32 accumulate, ~2^-22 per product).  The
                            packed weights are math-specific: pack and run with the same value. */
} dn_conv_desc;

/* floats needed for the packed weights of this conv */
size_t dn_conv_packed_weight_floats(const dn_conv_desc* d);
/* weight_oihw: [c_out][c0+c1][ksize][ksize] float32 (torch Conv2d layout; a
 * Conv3d (1,1,1) weight has the same bytes) -> packed tile-major layout. */
int dn_conv_pack_weights(const dn_conv_desc* d, const float* weight_oihw,
                         float* packed, void* stream);
/* scale/shift of y = relu?(acc * scale + shift) from conv bias and BatchNorm
 * running stats; gamma/beta/mean/var may all be NULL (no BN: scale=1,
 * shift=bias); bias may be NULL (0). */
int dn_fold_bn(const float* bias, const float* gamma, const float* beta,
               const float* mean, const float* var, float eps, int channels,
               float* scale, float* shift, void* stream);
int dn_conv2d(const dn_conv_desc* d, const float* src0, const float* src1,
              const float* packed, const float* scale, const float* shift,
              float* out, void* stream);
/* dn_conv2d of a 3x3 layer restricted to the taps of `tap_mask` (bit ky * 3 + kx; the others are skipped, not
 * multiplied by zero) with an explicitly strided output: pixel (oy, ox) of image n is written at
 * out + n * out_img_stride + oy * out_row_stride + ox * out_px_stride floats (+ channel).  The training step's
 * parity-phase stride-2 data gradient (disconet_train.h :: dn_conv_dgrad_class_weights) is four of these. */
int dn_conv2d_taps(const dn_conv_desc* d, const float* src0, const float* src1, const float* packed,
                   const float* scale, const float* shift, float* out, int tap_mask, long out_img_stride,
                   int out_row_stride, int out_px_stride, void* stream);

/* Fused "3x3 conv + affine + ReLU, then 1x1 conv + affine (+ReLU)" in one launch:
 * the activated 64-channel tile stays in LDS between the two layers.  Used for
 * the detection heads (conv1 of the cls and reg heads as one 64-channel conv,
 * their 1x1 conv2 as one block-diagonal second stage with two outputs) and for
 * conv*_2 + the (1,1,1) Conv3D of the encoder -- upstream ClassificationHead /
 * SingleRegressionHead / Backbone.encode.  Split-f16 math,
 * 3x3 stride 1, c_out == 64 only.  Columns [0, split) of the second stage go to
 * out_a (pixel stride ldo_a), columns [split, c_out2) to out_b (ldo_b). */
typedef struct dn_post1x1_desc {
  int32_t c_out2;        /* outputs of the 1x1 stage, multiple of 4, <= 64 */
  int32_t relu2;
  int32_t split;         /* multiple of 4; == c_out2 for a single output */
  int32_t ldo_a, ldo_b;
  int32_t block_diag;    /* dn_spconv2d_post1x1 only (dn_conv2d_post1x1 ignores it): the 1x1 stage is
                            block-diagonal -- outputs [0, split) read stage-1 channels 0..31, the rest
                            channels 32..63 (two detection heads side by side).  Needs out_f32, two
                            outputs, packed2 from dn_sp_post1x1_pack_heads. */
} dn_post1x1_desc;
size_t dn_post1x1_packed_floats(void);
/* w2: [c_out2][c_in2] float32, c_in2 <= 64 = channels of the first stage */
int dn_post1x1_pack_weights(const float* w2, int c_out2, int c_in2, float* packed, void* stream);
int dn_conv2d_post1x1(const dn_conv_desc* d, const dn_post1x1_desc* p, const float* src0,
                      const float* src1, const float* packed, const float* scale,
                      const float* shift, const float* packed2, const float* scale2,
                      const float* shift2, float* out_a, float* out_b, void* stream);

/* ------------------------------------------------------------------------
 * K2/K3/K7, split-planar ("SP") form -- the inference engine's conv path.
 * Same layers and same arithmetic as dn_conv2d with math = 1 (x = hi + lo f16
 * halves, hi*hi + hi*lo + lo*hi on the f16 MFMA, fp32 accumulate), but the
 * activations stay PRE-SPLIT in HBM and are staged by LDS-DMA, so the kernel's
 * loop is ds_read + MFMA only (disconet_amd/csrc/conv_sp.hip, sp_layout.h).
 *
 * SP tensor (opaque bytes, 16-byte aligned, dn_sp_tensor_bytes() long):
 *   [image][ceil(C/16) chunk][4 quarter][H][W] x 16 bytes; a piece = 8 halves =
 *   channels 16*chunk + 8*oct + 0..7 of one pixel; quarter = 2*part + oct,
 *   part 0 = half(x), part 1 = half(x - half(x)).  Channels past C are zero.
 * dn_conv_desc is reused: ld0/ld1/ldo are ignored; up0 is 0 or 1; math is ignored except
 * math == 3: source 0 is a HI-ONLY SP tensor (dn_scatter_dense_sp_hi; 3x3, stride 1, c1 == 0).
 * math == 4: source 0 is an occupancy BIT grid (dn_scatter_dense_bits; 3x3, stride 1, c1 == 0, c0 <= 32, c_out <= 32;
 * DN_ERR_UNSUPPORTED when the layer's weights do not fit the LDS); weights packed as for any other source.
 * Packed weights are specific to this engine (dn_spconv_pack_weights); `wmul`
 * is multiplied into the weights before the split -- pass a power of two that
 * lifts the layer's weights out of the f16 subnormal range and fold 1/wmul into
 * `scale` (exact in fp32).
 * The packed image depends on the layer's SOURCES as well as on its weights: pack
 * with the descriptor the layer will run with.  A 3x3 stride-1 layer whose first
 * source is nearest-upsampled (up0 = 1, c0 a multiple of 16, even h_in / w_in) is
 * packed TAP-MERGED: the kernel taps that read the same low-resolution pixel of
 * that source are summed per output-pixel parity class (16 blocks per 16-channel
 * chunk of source 0 -- 4 merged taps x 4 classes -- instead of 9) and dn_spconv2d
 * runs 2 x 2 taps over those chunks (disconet_amd/csrc/conv_spq.hip).
 * dn_spconv_packed_weight_bytes() accounts for it.
 * ------------------------------------------------------------------------ */
size_t dn_sp_tensor_bytes(int n_images, int h, int w, int channels);
/* fp32 NHWC [n][h][w][ld] (first `channels` of each pixel) <-> SP */
int dn_sp_from_nhwc(const float* src, int n_images, int h, int w, int channels, int ld,
                    void* dst_sp, void* stream);
int dn_sp_to_nhwc(const void* src_sp, int n_images, int h, int w, int channels, int ld,
                  float* dst, void* stream);
size_t dn_spconv_packed_weight_bytes(const dn_conv_desc* d);
int dn_spconv_pack_weights(const dn_conv_desc* d, const float* weight_oihw, float wmul,
                           void* packed, void* stream);
/* Many plain-layout packs in ONE launch (a training step packs every layer's weights once for its forward and once, flipped
 * and transposed, for its data gradient: ~70 launches of 4-7 us on a stream that has nothing else to run beside them).
 * A job packs, for the conv `desc`, the weight tensor W[c_out][c0 + c1][k][k] that a VIEW of `weight` defines:
 *   mode 0: W[n][ci][t] = weight[n][ci_first + ci][t], weight [c_out][cin_total][k][k]
 *           (= dn_spconv_pack_weights; of a column cut when cin_total > c0 + c1)
 *   mode 1: W[n][ci][t] = weight[ci][ci_first + n][k*k - 1 - t], weight [.][cin_total][k][k]
 *           (= dn_conv_dgrad_weights (disconet_train.h) then dn_spconv_pack_weights: the data gradient's conv)
 *   mode 2: W[cls n_in + j] = class (cls / 2, cls % 2) of dn_conv_dgrad_class_weights over column ci_first + j, desc.c_out = 4 n_in
 *           (the one-launch stride-2 data gradient: four classes as output-channel groups)
 * -- the same bytes as those calls write.  dn_spconv_pack_multi_prepare validates the jobs and fills the HOST image of the
 * device table (dn_spconv_pack_multi_table_bytes(n_jobs) bytes; DN_ERR_UNSUPPORTED for a layer that is packed tap-merged);
 * the caller copies it to the device once and calls dn_spconv_pack_weights_multi(table, n_jobs, total_blocks) whenever the
 * weights have changed (a job's wmul is part of the table). */
typedef struct dn_pack_job {
  dn_conv_desc desc;
  const float* weight;
  void* packed;
  int32_t mode, cin_total, ci_first, n_in;
  float wmul;
  int32_t reserved;
} dn_pack_job;
size_t dn_spconv_pack_multi_table_bytes(int n_jobs);
int dn_spconv_pack_multi_prepare(const dn_pack_job* jobs, int n_jobs, void* table_host, int* total_blocks);
int dn_spconv_pack_weights_multi(const void* table_device, int n_jobs, int total_blocks, void* stream);
/* The same for the fp32-NHWC engine's packs (dn_conv_pack_weights; desc.math == 1: the split-f16 rows): modes 0 and 1, and the
 * values are multiplied by the job's wmul first (a power of two: what a caller of dn_conv_pack_weights multiplies in itself). */
size_t dn_conv_pack_multi_table_bytes(int n_jobs);
int dn_conv_pack_multi_prepare(const dn_pack_job* jobs, int n_jobs, void* table_host, int* total_blocks);
int dn_conv_pack_weights_multi(const void* table_device, int n_jobs, int total_blocks, void* stream);
/* out: SP tensor [n_images][ceil(c_out/16)][4][h_out][w_out] */
int dn_spconv2d(const dn_conv_desc* d, const void* src0_sp, const void* src1_sp,
                const void* packed, const float* scale, const float* shift, void* out_sp,
                void* stream);
/* K-SLICED form of dn_spconv2d (3x3 layers; conv_sp.hip / conv_spq.hip `KSL` kernels).  `kslices` (1, 2 or 4) is a
 * property of the LAYER: every output is defined as the fp32 sum, in slice order and starting from zero, of
 * `kslices` accumulation chains over equal shares of the K loop (16-channel chunks; on the tap-merged up-conv equal
 * shares of its work).  How a launch distributes the slices does not change a bit of the result: a workgroup that
 * owns a whole tile folds them in registers, the tiles of the launch's last, under-filled round (every tile of a
 * launch smaller than the chip) are handed out slice by slice through `workspace` and added by a second, tiny launch
 * of the same kernel in the same order.  So the outputs of an image are bit-identical whatever the batch it is part
 * of (tests/test_gpu_conv.py), a 640-tile layer no longer runs two rounds on 512 resident workgroups, and the
 * 4-image launches of an agent-sharded rank fill the chip.  kslices = 1 is dn_spconv2d.  workspace may be NULL
 * (nothing is split) or smaller than dn_spconv_workspace_bytes() (fewer tiles are split).  Same packed weights as
 * dn_spconv2d.  Refused (DN_ERR_ARG): 1x1 layers, hi-only and bit-grid sources (math = 3, 4), the row-merged image
 * (dn_spconv_set_upmode(1)), layers with fewer chunks than slices. */
size_t dn_spconv_workspace_bytes(const dn_conv_desc* d, int kslices);
/* 1 if the layer can run with `kslices` canonical K slices (1, 2 or 4) in this process -- 3x3, no hi-only / bit-grid
 * source, not the row-merged up-conv image (DN_SP_UPMERGE=1), at least `kslices` 16-channel chunks -- else 0.  Depends on
 * the layer and the process-wide up-conv form only, never on the batch. */
int dn_spconv_ks_supported(const dn_conv_desc* d, int kslices);
int dn_spconv2d_ks(const dn_conv_desc* d, int kslices, const void* src0, const void* src1, const void* packed,
                   const float* scale, const float* shift, void* out, float* out_nhwc /* may be NULL: the second,
                   fp32 NHWC output of dn_spconv2d_dual */, int ld_nhwc,
                   void* workspace, size_t workspace_bytes, void* stream);

/* The encoder stem's first two layers in one launch (conv_pre_1 -> conv_pre_2, both 3x3 + BN + ReLU at the
 * full map): d1 / d2 describe the two layers (3x3, stride 1, one source each; c0 <= 16 -> 32 -> c_out <= 32, same images and
 * map), `bits` is the occupancy bit grid of dn_scatter_dense_bits, packed1 / packed2 the layers' dn_spconv_pack_weights
 * images, out the SP tensor of layer 2.  The intermediate map stays in LDS (it is never written); the result is
 * bit-identical to dn_spconv2d(d1 with math = 4) followed by dn_spconv2d(d2).  dn_spconv2d_pre_pair_supported: 1 if the pair
 * of descriptors fits this form (else DN_ERR_ARG from the launch). */
int dn_spconv2d_pre_pair_supported(const dn_conv_desc* d1, const dn_conv_desc* d2);
int dn_spconv2d_pre_pair(const dn_conv_desc* d1, const dn_conv_desc* d2, const uint32_t* bits, const void* packed1,
                         const float* scale1, const float* shift1, const void* packed2, const float* scale2,
                         const float* shift2, void* out_sp, void* stream);
/* The same conv with a SECOND copy of its output as float32 NHWC rows [n][h_out][w_out][ld_nhwc] (first c_out
 * columns; c_out % 4 == 0), written from the same epilogue registers before the f16 split: the level a
 * consumer outside the conv engine reads (the fusion kernels' maps, the agent all-gather) needs no
 * dn_sp_to_nhwc pass.  (Round 6: also on the tap-merged up-conv kernel, up0 = 1 layers.) */
int dn_spconv2d_dual(const dn_conv_desc* d, const void* src0_sp, const void* src1_sp, const void* packed,
                     const float* scale, const float* shift, void* out_sp, float* out_nhwc, int ld_nhwc,
                     void* stream);
/* dn_spconv2d whose ONLY output is the float32 NHWC copy (no SP tensor is written, nothing is split, no magnitude is
 * tracked): the training step's split-f16 data gradient -- src0 = dz as an SP tensor (dn_bn_train_backward_finish_sp),
 * packed = the flipped / transposed weights, scale = 1 / (sp_lift * wmul), shift = 0, relu = 0 -- and (round 6) the training
 * step's FORWARD convs: src = the previous layer's y as the SP tensor dn_bn_train_apply_mask_sp writes, scale = 1 / wmul,
 * shift = bias, out = z (what the BatchNorm statistics read).  Same restrictions as dn_spconv2d_dual. */
int dn_spconv2d_nhwc(const dn_conv_desc* d, const void* src0_sp, const void* src1_sp, const void* packed,
                     const float* scale, const float* shift, float* out_nhwc, int ld_nhwc, void* stream);
/* Fused 3x3 (64 channels) + affine + ReLU, then 1x1 + affine (+ReLU): the 64-channel tile
 * never leaves the registers between the two layers (cf. dn_conv2d_post1x1).
 * out_f32 == 0: out_a is an SP tensor of c_out2 channels (p->split, ldo_* ignored);
 * out_f32 != 0: fp32 NHWC, columns [0, split) -> out_a (ldo_a), the rest -> out_b (ldo_b). */
size_t dn_sp_post1x1_packed_bytes(void);
int dn_sp_post1x1_pack_weights(const float* w2, int c_out2, int c_in2, float wmul, void* packed,
                               void* stream);
/* w2 [c_out2][64] of a block-diagonal stage (see dn_post1x1_desc.block_diag); dn_sp_post1x1_packed_bytes() */
int dn_sp_post1x1_pack_heads(const float* w2, int c_out2, int split, float wmul, void* packed,
                             void* stream);
int dn_spconv2d_post1x1(const dn_conv_desc* d, const dn_post1x1_desc* p, const void* src0_sp,
                        const void* src1_sp, const void* packed, const float* scale,
                        const float* shift, const void* packed2, const float* scale2,
                        const float* shift2, int out_f32, void* out_a, float* out_b, void* stream);
/* tools only: force tile configuration `cfg` (an index of conv_sp.hip's menu) where it
 * applies to the layer, -1 = automatic selection.  Process-wide, not thread-safe. */
int dn_spconv_force_config(int cfg);
/* tools only: form of the packed image / kernel of layers whose first source is upsampled: 0 = plain taps,
 * 1 = row-merged, 2 = row- and column-merged per parity class (default), -1 = the DN_SP_UPMERGE
 * environment value.  Process-wide; weights packed under one mode must run under the same mode. */
int dn_spconv_set_upmode(int mode);

/* ------------------------------------------------------------------------
 * K4 -- pose-based two-pass bilinear warp of neighbour feature maps.
 * Replaces upstream:coperception/models/det/base/* :: feature_transformation
 * (+ build_neighbors_feature_list) (Appx A.4): rotate
 * (affine_grid + grid_sample, bilinear, zeros, align_corners=False), zero-pad,
 * then translate by (4*t_x/128, -4*t_y/128) in normalised units.
 *   feat   [A*B][H][W][C]  agent-major layer-`layer` maps
 *   trans  [B][A][A][4][4] float32,  trans[b][i][j] maps j -> i
 *   num_agent [B] int32 live-agent count per sample
 *   ego_first, ego_count: the egos i in [ego_first, ego_first + ego_count) this
 *          call serves (all of them on one GPU: 0, A; one agent per GPU when the
 *          agents of a scene are sharded across ranks).
 *   warped [B][ego_count][A-1][H][W][C]; slot (b, i - ego_first, jj),
 *          jj = j - (j > i), receives warp(j -> i); slots of dead agents are
 *          zero-filled.
 *   only_v2i != 0: only pairs with i == 0 or j == 0 are warped (others zero).
 * ------------------------------------------------------------------------ */
int dn_warp_neighbors(const float* feat, const float* trans, const int32_t* num_agent,
                      int batch, int agents, int h, int w, int c, int only_v2i,
                      int ego_first, int ego_count, float* warped, void* stream);

/* ------------------------------------------------------------------------
 * K5 tail + K6 -- per-pixel attention MLP tail, softmax over agents, weighted
 * sum.  Replaces PixelWeightedFusionSoftmax.forward layers 2-4 and the fusion
 * loop body of upstream:coperception/models/det/DiscoNet.py :: DiscoNet.forward
 * (Appx A.5).  Layer 1 (1x1 conv 2C -> 128) is split as
 * W1 = [W1_ego | W1_nbr] and evaluated with dn_conv2d:
 *   g      [ego_count*B][H*W][256] = [ x.W1_ego^T + b1 | x.W1_nbr^T ] of the served
 *          egos' own maps (image = (i - ego_first) * B + b)
 *   fw     [B][ego_count][A-1][H*W][128] = warped . W1_nbr^T
 *   feat   [A*B][H*W][C] maps of ALL agents; fused [ego_count*B][H*W][C]
 * The tail computes, per ego i < num_agent[b], per pixel, for neighbours
 * k = ego, then j ascending (j != i):
 *   h1 = relu(bn1(E + F_k)); h2 = relu(bn2(W2 h1 + b2)); h3 = relu(bn3(W3 h2 + b3));
 *   s_k = relu(W4 h3 + b4); w_k = exp(s_k) / sum exp(s_.)   (no max-shift)
 *   fused = sum_k w_k * nbr_k
 * and copies feat for dead agents.  mlp params (device, float32):
 *   bn1_scale/shift[128] (BN affine only, bias b1 is already in E),
 *   w2[32][128], s2/t2[32] (bias+BN folded), w3[8][32], s3/t3[8], w4[8], b4[1].
 * ------------------------------------------------------------------------ */
typedef struct dn_mlp_tail_params {
  const float* bn1_scale; const float* bn1_shift;
  const float* w2; const float* s2; const float* t2;
  const float* w3; const float* s3; const float* t3;
  const float* w4; const float* b4;
} dn_mlp_tail_params;

int dn_disco_fuse_tail(const float* feat, const float* warped, const float* g,
                       const float* fw, const int32_t* num_agent,
                       const dn_mlp_tail_params* p, int batch, int agents, int hw, int c,
                       int only_v2i, int ego_first, int ego_count, float* fused,
                       float* weights_out, /* <- may be NULL; [B][ego_count][A][hw] softmax
                       weights in neighbour-list order */ void* stream);

/* ------------------------------------------------------------------------
 * K5 + K6 in ONE launch (disconet_amd/csrc/fuse_mlp.hip): all four layers of the pairwise
 * attention MLP on the f16 MFMA (split-f16 x3, fp32 accumulate), exp / sum over the agents
 * and the weighted sum, lanes over pixels, no intermediate tensors.  Same semantics, inputs
 * (feat, warped, num_agent, ego range, only_v2i) and neighbour order as dn_disco_fuse_tail;
 * replaces the dn_conv2d (layer 1) + dn_disco_fuse_tail pair for c in {64, 128, 256}.
 *   packed: dn_fuse_mlp_pack() of conv1_1.weight [128][2c] (= [W_ego | W_nbr]),
 *           conv1_2.weight [32][128], conv1_3.weight [8][32]; each matrix is multiplied by
 *           its wmul (a power of two, see dn_spconv_pack_weights) before the f16 split;
 *   s1/t1[128]: y = relu(acc * s1 + t1) after layer 1 (bias, BN and 1/wmul1 folded),
 *   s2/t2[32], s3/t3[8] likewise; w4[8], b4[1] of the last layer (fp32 dot, ReLU).
 *   fused_sp (SP tensor [ego_count*batch][c/16][4][hw] x 16 B) and/or fused_nhwc
 *   ([ego_count*batch][hw][c] float32); weights_out as dn_disco_fuse_tail.
 * ------------------------------------------------------------------------ */
typedef struct dn_fuse_mlp_params {
  const void* packed;
  const float* s1; const float* t1;
  const float* s2; const float* t2;
  const float* s3; const float* t3;
  const float* w4; const float* b4;
} dn_fuse_mlp_params;
int dn_fuse_mlp_supported(int c);
/* tools / tests only: the launch form of dn_disco_fuse_mlp -- 4 (the ego term, the list slots and the channels of the
 * weighted sum of a 32-pixel tile split over four waves: launches of fewer than 512 tiles), 2 (round 5: one wave per
 * tile, workgroups of 2-4 tiles that stage the layer-1 weights in LDS once: 512 tiles and more) or 1 (one wave per
 * tile streaming its weight fragments from L2: rounds 2-4's form for large launches, kept for A/B);
 * 0 = chosen per launch unless DN_FUSE_MLP_WAVES is set.  All forms give bit-identical results.
 * Process-wide, not thread-safe. */
int dn_fuse_mlp_set_waves(int waves);
size_t dn_fuse_mlp_packed_bytes(int c);
int dn_fuse_mlp_pack(const float* w1, const float* w2, const float* w3, int c, float wmul1,
                     float wmul2, float wmul3, void* packed, void* stream);
int dn_disco_fuse_mlp(const float* feat, const float* warped, const int32_t* num_agent,
                      const dn_fuse_mlp_params* p, int batch, int agents, int hw, int c,
                      int only_v2i, int ego_first, int ego_count, void* fused_sp,
                      float* fused_nhwc, float* weights_out, void* stream);
/* FRAGMENT-MAJOR form of the warped neighbour maps (the default of the Python host when h * w % 32 == 0 and
 * c % 64 == 0).  dn_warp_neighbors_fm writes each (sample, ego, neighbour) block of hw * c floats as
 *   [tile t of 32 pixels][k-step ks of 16 channels][half r][lane = 32 h + j] x 4 floats
 *   = channels 16 ks + 8 h + 4 r + 0..3 of pixel 32 t + j
 * -- the order dn_disco_fuse_mlp_fm's wavefronts read it (lane (j, h) of the wave that owns tile t holds exactly these
 * pieces as its MFMA operand), so that every load instruction of a wave is one contiguous 1 KB run instead of 32 half
 * cache lines.  Same values, same arithmetic and same results as dn_warp_neighbors + dn_disco_fuse_mlp; the block is an
 * intermediate between the two calls, nothing else reads it. */
int dn_warp_fm_supported(int h, int w, int c);
int dn_warp_neighbors_fm(const float* feat, const float* trans, const int32_t* num_agent,
                         int batch, int agents, int h, int w, int c, int only_v2i,
                         int ego_first, int ego_count, float* warped_fm, void* stream);
int dn_disco_fuse_mlp_fm(const float* feat, const float* warped_fm, const int32_t* num_agent,
                         const dn_fuse_mlp_params* p, int batch, int agents, int hw, int c,
                         int only_v2i, int ego_first, int ego_count, void* fused_sp,
                         float* fused_nhwc, float* weights_out, void* stream);


/* ------------------------------------------------------------------------
 * Detection decode (first step after the hot path).
 * Replaces the dense part of upstream:coperception/utils/postprocess.py that
 * CoDetModule.predict_all runs on the CPU: foreground probability = softmax of
 * the 2 class logits, box = anchor-relative decode of the 6-value code
 * (x, y, w, h, sin, cos).  NMS / mAP stay on the CPU as in the reference.
 *   cls [n_images][anchors_per_image][2], loc [n_images][anchors_per_image][6],
 *   anchors [anchors_per_image][6] -> scores [n][apl], boxes [n][apl][6]
 * ------------------------------------------------------------------------ */
int dn_decode_boxes(const float* cls, const float* loc, const float* anchors, int n_images,
                    long anchors_per_image, float* scores, float* boxes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DISCONET_HIP_H */
