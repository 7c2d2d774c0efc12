// Implicit-GEMM convolution for gfx950 on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32), NHWC activations, fused affine (+ReLU) epilogue,
// optional on-the-fly "nearest x2 upsample of src0, concat src1" input.
//
// GEMM view: M = output pixels (a TH x TW spatial tile per workgroup),
// N = output channels (BN per workgroup), K = taps x input channels.
// Per input-channel chunk (KC channels) the workgroup stages ONE halo patch
// [(TH-1)*S+KS][(TW-1)*S+KS][KC] and the matching weights [taps][BN][KC] in
// LDS; the KS*KS taps then read the same patch at shifted pixel offsets, so
// the input is fetched once per chunk instead of once per tap (no im2col
// materialisation, no 9x re-read).
//
// Software pipeline: the global loads of chunk c+1 are issued into registers
// BEFORE the MFMA block of chunk c and written to LDS after it, so HBM/L2
// latency hides under the 64-cycle MFMAs; 2-3 workgroups per CU cover each
// other's barrier bubbles.  Staging uses buffer loads: the per-lane byte
// offsets of a tile are computed ONCE, a chunk only changes the scalar offset,
// and halo / tail slots carry an out-of-range offset so the hardware bounds
// check returns the zero padding (no per-chunk address math, clamps or selects
// -- measured as ~20 % of the kernel when done on the VALU).
//
// Orientation: the MFMA's M dimension carries OUTPUT CHANNELS and N carries
// pixels (D[i = channel][j = pixel]); lane l then owns one pixel (l&31) and, per
// register quad, 4 consecutive channels -> the epilogue is 16-byte stores
// (4 per 32x32 tile instead of 16 dword stores; the dword form cost ~25 % of the
// 32-channel full-resolution layers).
//
// MFMA operand mapping (cdna_hip_programming.md §3): for 32x32x2 lane l holds
// A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].  Both LDS images keep K
// contiguous per row, so one ds_read_b128 per operand feeds four MFMAs: lane
// (i, h) reads k = 8s + 4h + {0..3}; the K order inside the chunk is a
// permutation shared by A and B, which only changes the fp32 summation order.
// Row stride KC+4 floats keeps every 16-lane ds_read_b128 group on 16
// distinct 16-byte slots (conflict-free for stride 1).
//
// Math modes (dn_conv_desc.math):
//   0  exact fp32: v_mfma_f32_32x32x2_f32, bit-compatible with an fp32 fmaf chain
//      (157 TFLOP/s peak = 1/16 of the f16/bf16 MFMA rate).
//   1  split-f16: every fp32 operand x is split x = hi + lo with hi = half(x),
//      lo = half(x - hi) (22-bit significand together; fp16 subnormals are kept by
//      the MFMA, probed in tools/mfma_f16_probe.hip) and a product is evaluated as
//      hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation:
//      3 MFMAs of 32 cycles per 16 k instead of 8 of 64 cycles -> 5.3x the MFMA
//      throughput at an error of ~2^-22 per product (the dropped lo*lo term),
//      i.e. the same class as fp32 summation-order noise (whole-network max abs
//      error 1e-5 on O(1) activations vs the 1e-4 parity bar).  Activations stay
//      fp32 in HBM; weights are pre-split at pack time into [hi halves | lo halves]
//      rows of the same byte size.
//
// Packed weights are tile-independent: [chunk of KCP cin][tap][c_out padded to
// 32][KCP], KCP = 16 for 3x3 and 32 for 1x1; a tile with KC | KCP reads
// sub-rows, so the tile shape can be chosen per launch from the problem size.
//
// Replaces the conv2d/conv3d(1x1x1)+batch_norm+relu(+interpolate+cat) chains of
// upstream:coperception/models/det/backbone/Backbone.py.
#include "dn_internal.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

namespace {

struct TileCoord {
  int img, oy0, ox0, n0;
};

struct ConvArgs {
  const float* src0;
  const float* src1;
  const float* wpk;
  const float* scale;
  const float* shift;
  float* out;
  int n_images, h_in, w_in, h_out, w_out;
  int c0, c1, up0, c_out, relu;
  int ld0, ld1, ldo;
  int nchunks, tiles_x, tiles_y;
  int vec0, vec1;
  int cout_pad;   // rows per (chunk, tap) in the packed weights
  int wpk_bytes;  // size of the packed weights (buffer bounds)
  int total_items;     // work items = channel blocks x images x tiles
  int vec_out;    // out / scale / shift allow 16-byte accesses
  int xcd_order;  // work-item order keeps a pixel tile's channel blocks on one XCD
  // fused 1x1 stage (POST kernels only): out2 = act2(h . W2^T * scale2 + shift2)
  const float* w2;      // packed split-f16 rows [64][BN hi halves | BN lo halves]
  const float* scale2;
  const float* shift2;
  float* out_b;         // columns [split2, c_out2) of the fused stage (may be null)
  int c_out2, relu2, split2, ldo_b;
  // dn_conv2d_taps: only the taps of `tap_mask` are multiplied (bit ky * 3 + kx), and the output pixel (oy, ox) of
  // image n lands at out + n * out_img + oy * out_row + ox * out_px floats (POST == 0 epilogue)
  int tap_mask;
  long out_img;
  int out_row, out_px;
};

constexpr int kcp_of(int ksize) { return ksize == 3 ? 16 : 32; }
constexpr int kNumCUs = 256;   // MI355X

template <int KS, int STRIDE, int TH, int TW, int BN, int KC, int WAVES_M, int WAVES_N,
          int WTM, int WTN>
struct ConvTile {
  static constexpr int NT = WAVES_M * WAVES_N * 64;
  static constexpr int BM = TH * TW;
  static constexpr int PAD = KS / 2;
  static constexpr int PH = (TH - 1) * STRIDE + KS;
  static constexpr int PW = (TW - 1) * STRIDE + KS;
  static constexpr int PS = KC + 4;
  static constexpr int TAPS = KS * KS;
  static constexpr int KV = KC / 4;
  static constexpr int KCP = kcp_of(KS);
  static constexpr int A_VEC = PH * PW * KV;            // float4 slots of the patch
  static constexpr int B_VEC = TAPS * BN * KV;          // float4 slots of the weights
  static constexpr int A_IT = (A_VEC + NT - 1) / NT;
  static constexpr int B_IT = (B_VEC + NT - 1) / NT;
  static constexpr int A_FLOATS = PH * PW * PS;
  static constexpr int B_FLOATS = TAPS * BN * PS;
  static constexpr int CS = BN + 4;                      // epilogue staging row stride
  static constexpr int C_FLOATS = BM * CS;
  static constexpr size_t LDS_BYTES =
      (size_t)(A_FLOATS + B_FLOATS > C_FLOATS ? A_FLOATS + B_FLOATS : C_FLOATS) * sizeof(float);
  // workgroups per CU the 160 KiB LDS admits; the kernel is register-bounded to match
  static constexpr int OCC_LDS = (int)(160 * 1024 / LDS_BYTES);
  static constexpr int OCC = OCC_LDS < 2 ? 2 : (OCC_LDS > 4 ? 4 : OCC_LDS);
  static_assert(BM == WAVES_M * WTM * 32, "pixel tile must match the wave layout");
  static_assert(BN == WAVES_N * WTN * 32, "channel tile must match the wave layout");
  static_assert(KC % 8 == 0 && KCP % KC == 0, "KC must cover b128 operand pairs and divide KCP");
};

// ABL != 0 builds ablation variants for tools/conv_ablate.hip only (the product
// always launches ABL = 0): 1 = no steady-state global loads / LDS stores /
// barriers, 3 = no epilogue stores, 5 = 1 + 3 + MFMA operands from registers
// (pure MFMA stream), 8 = split-f16 staging without the fp32 -> hi/lo conversion
// (what a pre-split activation format would cost).
template <int KS, int STRIDE, int TH, int TW, int BN, int KC, int WAVES_M, int WAVES_N,
          int WTM, int WTN, int ABL = 0, int MATH = 0, int POST = 0>
__global__ void __launch_bounds__(
    (WAVES_M * WAVES_N * 64),
    (ConvTile<KS, STRIDE, TH, TW, BN, KC, WAVES_M, WAVES_N, WTM, WTN>::OCC))
conv_mfma_kernel(const ConvArgs a) {
  constexpr bool kSplit = MATH == 1;
  // POST: a 1x1 conv (<= 64 outputs) on this layer's activated output, fused into
  // the epilogue: the tile never leaves the CU between the two layers (heads:
  // conv1 + conv2 of cls and reg in one launch; conv*_2 + the 1x1x1 Conv3D).
  static_assert(POST == 0 || (kSplit && WTN == 1 && BN == 64 && WAVES_N == 2),
                "fused 1x1 stage: split-f16, 64-channel tile, 2 channel waves");
  constexpr bool kNoStream = ABL == 1 || ABL == 5;
  constexpr bool kNoStore = ABL == 3 || ABL == 5;
  constexpr bool kNoLds = ABL == 5;
  constexpr bool kNoBStage = ABL == 9;    // steady state without weight staging
  constexpr bool kNoAStage = ABL == 10;   // steady state without activation staging
  using T = ConvTile<KS, STRIDE, TH, TW, BN, KC, WAVES_M, WAVES_N, WTM, WTN>;
  constexpr int NT = T::NT, PW = T::PW, PS = T::PS, TAPS = T::TAPS, KV = T::KV, KCP = T::KCP;
  constexpr int ROWS_PER_IT = NT / KV;   // weight rows one staging iteration covers
  static_assert(TAPS == 1 || ROWS_PER_IT % BN == 0, "weight staging must step whole taps");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + T::A_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;
  const int li = lane & 31;
  const int lh = lane >> 5;

  // Work items = (channel block, image, tile_y, tile_x), tile_x fastest.  A launch
  // gives every item its own workgroup (gridDim.x == total_items) or runs
  // persistent workgroups that walk items blockIdx.x, + gridDim.x, ... and fetch
  // the first chunk of their next tile under the MFMAs of the current tile's last
  // chunk.  All workgroups of a launch start together and take the same time, so
  // one-workgroup-per-item launches hit HBM with every prologue load of a
  // "generation" at once (measured ~30 % of the 2-chunk layers); the persistent
  // form spreads those loads under the MFMA stream.
  const int G = gridDim.x;
  const int spatial_items = a.n_images * a.tiles_y * a.tiles_x;
  // XCD-aware order (a.xcd_order): workgroup b runs on XCD b % 8 (observed, not a
  // contract -- a wrong guess only costs speed), each XCD has its own L2.  All
  // channel blocks of a pixel tile re-read the same patch, so they take ids
  // b, b + 8, b + 16, ...: same XCD, dispatched back to back, the patch comes out of
  // that L2 instead of HBM once per channel block.
  const int n_cb = a.total_items / spatial_items;
  const int sp_full = spatial_items & ~7;
  auto decode = [&](int item) {
    TileCoord tc;
    int cb, sp;
    if (!a.xcd_order) {
      cb = item / spatial_items;
      sp = item % spatial_items;
    } else if (item < sp_full * n_cb) {
      const int j = item >> 3;
      cb = j % n_cb;
      // order 2: each XCD walks its own contiguous run of pixel tiles, so tiles that
      // share halo rows / columns meet in one L2 as well
      sp = a.xcd_order == 2 ? (item & 7) * (sp_full >> 3) + j / n_cb : (j / n_cb) * 8 + (item & 7);
    } else {
      const int r = item - sp_full * n_cb, rem = spatial_items - sp_full;
      cb = r / rem;
      sp = sp_full + r % rem;
    }
    tc.n0 = cb * BN;
    tc.ox0 = (sp % a.tiles_x) * TW;
    sp /= a.tiles_x;
    tc.oy0 = (sp % a.tiles_y) * TH;
    tc.img = sp / a.tiles_y;
    return tc;
  };

  int a_off[WTM], b_off[WTN];
#pragma unroll
  for (int wm = 0; wm < WTM; ++wm) {
    const int m = (wave_m * WTM + wm) * 32 + li;
    a_off[wm] = ((m / TW) * STRIDE * PW + (m % TW) * STRIDE) * PS + ((kSplit && KC == 8) ? 2 : 4) * lh;
  }
#pragma unroll
  for (int wn = 0; wn < WTN; ++wn) {
    const int n = (wave_n * WTN + wn) * 32 + li;
    b_off[wn] = n * PS + ((kSplit && KC == 8) ? 2 : 4) * lh;
  }

  f32x16 acc[WTM][WTN];
  const f32x4 abl_const = {li * 1e-3f, 0.5f, -0.25f, lh * 1.f};   // ablation operands only
  f32x4 ra[T::A_IT], rb[T::B_IT];   // the next chunk, in flight from global memory
  // fused stage's weights: this lane's B-operand fragments of W2 (row n2 = wave_n * 32 + lane % 32,
  // k-slices by lane / 32) do not depend on the tile -- 32 VGPRs held for the whole launch instead
  // of an LDS image rewritten every tile
  half8 w2h[POST ? BN / 16 : 1], w2l[POST ? BN / 16 : 1];
  if constexpr (POST != 0) {
    const float* row = a.w2 + (size_t)(wave_n * 32 + li) * BN;
#pragma unroll
    for (int s2 = 0; s2 < BN / 16; ++s2) {
      w2h[s2] = *reinterpret_cast<const half8*>(row + 8 * s2 + 4 * lh);
      w2l[s2] = *reinterpret_cast<const half8*>(row + BN / 2 + 8 * s2 + 4 * lh);
    }
  }


  // ---- staging state.  Buffer loads: per-lane byte offsets are computed once per
  // tile (and once more where a concat layer switches source); a chunk only moves
  // the scalar offset.  Halo / tail slots carry an out-of-range offset, so the
  // hardware bounds check supplies the zero padding.
  constexpr unsigned OOB = 0xFFFFFFFFu;   // >= num_records: the load returns 0
  const int hs0 = a.up0 ? (a.h_in >> 1) : a.h_in, ws0 = a.up0 ? (a.w_in >> 1) : a.w_in;
  const size_t img0_floats = (size_t)hs0 * ws0 * a.ld0, img1_floats = (size_t)a.h_in * a.w_in * a.ld1;
  auto rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0), 0, 0, 0x00020000);
  auto rsrc1 = rsrc0;
  const auto rsrcw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk), 0, a.wpk_bytes,
                                                        0x00020000);
  unsigned voff_a[T::A_IT];   // patch slots of the source being read
  unsigned voff_b;            // first weight row of this lane; later rows via the scalar offset
  // weights of one staging iteration further on: whole taps for KS = 3, rows for 1x1
  const int b_step = (TAPS == 1 ? ROWS_PER_IT : (ROWS_PER_IT / BN) * a.cout_pad) * KCP * 4;
  const int src_switch = a.c1 > 0 ? a.c0 / KC : (1 << 30);   // first chunk read from src1

  // The per-lane values below depend on the tile only through scalars; deriving
  // them from an opaque copy of tid keeps hipcc from hoisting them out of the tile
  // loop and holding ~40 extra VGPRs live across it.
  auto opaque_tid = [&]() {
    int t = tid;
    asm volatile("" : "+v"(t));
    return t;
  };
  auto setup_voff_a = [&](const TileCoord& tc, bool from1) {
    const int t = opaque_tid();
    const int iy0 = tc.oy0 * STRIDE - T::PAD, ix0 = tc.ox0 * STRIDE - T::PAD;
#pragma unroll
    for (int it = 0; it < T::A_IT; ++it) {
      const int idx = t + it * NT;
      const int p = idx / KV, q = idx % KV;
      const int iy = iy0 + p / PW, ix = ix0 + p % PW;
      bool ok = idx < T::A_VEC && iy >= 0 && iy < a.h_in && ix >= 0 && ix < a.w_in;
      // up0 == 2: source 0 is read zero-stuffed (data gradient of a stride-2 layer)
      if (!from1 && a.up0 == 2) ok = ok && !((iy | ix) & 1);
      unsigned off;
      if (from1) {
        off = (unsigned)(((iy * a.w_in + ix) * a.ld1 + 4 * q) * 4);
      } else {
        const int sy = a.up0 ? (iy >> 1) : iy, sx = a.up0 ? (ix >> 1) : ix;
        off = (unsigned)(((sy * ws0 + sx) * a.ld0 + 4 * q) * 4);
      }
      voff_a[it] = ok ? off : OOB;
    }
  };
  auto setup_tile = [&](const TileCoord& tc) {
    rsrc0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.src0 + tc.img * img0_floats), 0, (int)(img0_floats * 4), 0x00020000);
    rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.c1 ? a.src1 + tc.img * img1_floats : a.src0), 0,
        a.c1 ? (int)(img1_floats * 4) : 0, 0x00020000);
    setup_voff_a(tc, false);
    const int t = opaque_tid();
    const int row = t / KV, q = t % KV;           // row = tap * BN + n
    // rows past c_out's padded count read neighbouring weights: finite values that
    // only reach output channels the epilogue never stores
    // split-f16 rows are [KCP hi halves | KCP lo halves]; a KC-wide tile takes the 16-byte
    // piece q from the hi part (q < KC/8) or from the lo part, linear when KC == KCP
    const int piece = (kSplit && KC < KCP) ? (q < KC / 8 ? 4 * q : KCP / 2 + 4 * (q - KC / 8)) : 4 * q;
    voff_b = (unsigned)(((((row / BN) * a.cout_pad + tc.n0 + row % BN) * KCP) + piece) * 4);
  };

  auto ld128 = [](auto rsrc, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
  };
  auto ld32x4 = [](auto rsrc, unsigned voff, int soff) {
    // a channel count that is not a multiple of 4 (the 13-bin voxel input): dword
    // loads.  Channels past the source's count read the neighbouring pixel (or 0
    // past the image) and meet zero weights.
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                           rsrc, voff == 0xFFFFFFFFu ? voff : voff + 4 * e, soff, 0));
    return v;
  };

  // issue the global loads of chunk `ch` of tile `tc` into ra/rb
  auto load_chunk = [&](const TileCoord& tc, int ch) {
    const int cbeg = ch * KC;
    if (ch == src_switch) setup_voff_a(tc, true);
    if (kNoAStage && ch > 0) {
    } else if (cbeg < a.c0) {
      const int soff = cbeg * 4;
      if (a.vec0) {
#pragma unroll
        for (int it = 0; it < T::A_IT; ++it) ra[it] = ld128(rsrc0, voff_a[it], soff);
      } else {
#pragma unroll
        for (int it = 0; it < T::A_IT; ++it) ra[it] = ld32x4(rsrc0, voff_a[it], soff);
      }
    } else {
      const int soff = (cbeg - a.c0) * 4;
      if (a.vec1) {
#pragma unroll
        for (int it = 0; it < T::A_IT; ++it) ra[it] = ld128(rsrc1, voff_a[it], soff);
      } else {
#pragma unroll
        for (int it = 0; it < T::A_IT; ++it) ra[it] = ld32x4(rsrc1, voff_a[it], soff);
      }
    }
    // weights: packed [chunk KCP][tap][cout_pad][KCP]; this tile reads the KC-wide
    // sub-row (ch % (KCP/KC)) of rows n0..n0+BN
    // sub-chunk inside a packed chunk: KC floats on for fp32 rows, KC halves on for split rows
    const int wsoff = (cbeg / KCP) * TAPS * a.cout_pad * KCP * 4 + (cbeg % KCP) * (kSplit ? 2 : 4);
    if (!(kNoBStage && ch > 0)) {
#pragma unroll
      for (int it = 0; it < T::B_IT; ++it) rb[it] = ld128(rsrcw, voff_b, wsoff + it * b_step);
    }
  };

  auto store_chunk = [&](bool steady = false) {
#pragma unroll
    for (int it = 0; it < (kNoAStage && steady ? 0 : T::A_IT); ++it) {
      const int idx = tid + it * NT;
      if (kSplit && ABL == 8) {   // ablation: staging without the fp32 -> hi/lo split
        if (idx < T::A_VEC) {
          float* rowp = &As[(idx / KV) * PS];
          *reinterpret_cast<f32x2*>(rowp + 2 * (idx % KV)) = f32x2{ra[it][0], ra[it][1]};
          *reinterpret_cast<f32x2*>(rowp + KC / 2 + 2 * (idx % KV)) = f32x2{ra[it][2], ra[it][3]};
        }
      } else if (kSplit) {
        // x = hi + lo, hi = half(x) (clamped to the finite fp16 range), lo = half(x - hi);
        // row layout [KC hi halves | KC lo halves]: slot q owns halves 4q..4q+3 of each part
        half4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = fminf(fmaxf(ra[it][e], -65504.f), 65504.f);
          hi[e] = (_Float16)x;
          lo[e] = (_Float16)(x - (float)hi[e]);
        }
        if (idx < T::A_VEC) {
          float* rowp = &As[(idx / KV) * PS];
          *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(rowp) + 4 * (idx % KV)) = hi;
          *reinterpret_cast<half4*>(reinterpret_cast<_Float16*>(rowp) + KC + 4 * (idx % KV)) = lo;
        }
      } else {
        if (idx < T::A_VEC) *reinterpret_cast<f32x4*>(&As[(idx / KV) * PS + 4 * (idx % KV)]) = ra[it];
      }
    }
#pragma unroll
    for (int it = 0; it < (kNoBStage && steady ? 0 : T::B_IT); ++it) {
      const int idx = tid + it * NT;
      if (idx < T::B_VEC) *reinterpret_cast<f32x4*>(&Bs[(idx / KV) * PS + 4 * (idx % KV)]) = rb[it];
    }
  };

  auto mfma_chunk = [&]() {
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      if (TAPS > 1 && !((a.tap_mask >> tap) & 1)) continue;      // uniform: a masked tap costs one scalar test
      const int toff = ((tap / KS) * PW + (tap % KS)) * PS;
      if constexpr (!kSplit) {
#pragma unroll
        for (int s = 0; s < KC / 8; ++s) {
          f32x4 av[WTM], bv[WTN];
#pragma unroll
          for (int wm = 0; wm < WTM; ++wm)
            av[wm] = kNoLds ? abl_const : *reinterpret_cast<const f32x4*>(&As[a_off[wm] + toff + 8 * s]);
#pragma unroll
          for (int wn = 0; wn < WTN; ++wn)
            bv[wn] = kNoLds ? abl_const
                            : *reinterpret_cast<const f32x4*>(&Bs[tap * BN * PS + b_off[wn] + 8 * s]);
          // D[i = channel][j = pixel]: weights are the MFMA's A operand, pixels its B
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int wm = 0; wm < WTM; ++wm)
#pragma unroll
              for (int wn = 0; wn < WTN; ++wn)
                acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[wn][t], av[wm][t],
                                                                    acc[wm][wn], 0, 0, 0);
        }
      } else if constexpr (KC >= 16) {
        // split-f16, 16 k per MFMA: lane half g reads halves 16s+8g..+7 of the hi part
        // and the same of the lo part (the k <-> slot map is shared by both operands)
#pragma unroll
        for (int s = 0; s < KC / 16; ++s) {
          half8 ah[WTM], al[WTM], bh[WTN], bl[WTN];
#pragma unroll
          for (int wm = 0; wm < WTM; ++wm) {
            ah[wm] = *reinterpret_cast<const half8*>(&As[a_off[wm] + toff + 8 * s]);
            al[wm] = *reinterpret_cast<const half8*>(&As[a_off[wm] + toff + KC / 2 + 8 * s]);
          }
#pragma unroll
          for (int wn = 0; wn < WTN; ++wn) {
            bh[wn] = *reinterpret_cast<const half8*>(&Bs[tap * BN * PS + b_off[wn] + 8 * s]);
            bl[wn] = *reinterpret_cast<const half8*>(&Bs[tap * BN * PS + b_off[wn] + KC / 2 + 8 * s]);
          }
#pragma unroll
          for (int wm = 0; wm < WTM; ++wm)
#pragma unroll
            for (int wn = 0; wn < WTN; ++wn) {
              acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[wn], ah[wm], acc[wm][wn], 0, 0, 0);
              acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[wn], al[wm], acc[wm][wn], 0, 0, 0);
              acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[wn], ah[wm], acc[wm][wn], 0, 0, 0);
            }
        }
      } else {
        // split-f16 with 8-channel chunks: the 8-k MFMA form, 4 halves per lane half
        half4 ah[WTM], al[WTM], bh[WTN], bl[WTN];
#pragma unroll
        for (int wm = 0; wm < WTM; ++wm) {
          ah[wm] = *reinterpret_cast<const half4*>(&As[a_off[wm] + toff]);
          al[wm] = *reinterpret_cast<const half4*>(&As[a_off[wm] + toff + KC / 2]);
        }
#pragma unroll
        for (int wn = 0; wn < WTN; ++wn) {
          bh[wn] = *reinterpret_cast<const half4*>(&Bs[tap * BN * PS + b_off[wn]]);
          bl[wn] = *reinterpret_cast<const half4*>(&Bs[tap * BN * PS + b_off[wn] + KC / 2]);
        }
#pragma unroll
        for (int wm = 0; wm < WTM; ++wm)
#pragma unroll
          for (int wn = 0; wn < WTN; ++wn) {
            acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x8f16(bl[wn], ah[wm], acc[wm][wn], 0, 0, 0);
            acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x8f16(bh[wn], al[wm], acc[wm][wn], 0, 0, 0);
            acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x8f16(bh[wn], ah[wm], acc[wm][wn], 0, 0, 0);
          }
      }
    }
  };

  // Epilogue.  C/D layout: column = lane&31 = pixel, row = (r&3) + 8*(r>>2) +
  // 4*(lane>>5) = channel, so registers 4g..4g+3 are channels 8g+4h+{0..3} of this
  // lane's pixel.  The affine+ReLU'd tile is staged through LDS as [pixel][BN] and
  // written out row-major: consecutive lanes cover consecutive 16-byte pieces of a
  // pixel's channels and consecutive pixels of a tile row are contiguous in NHWC,
  // so every store instruction is one contiguous run.
  constexpr int CS = T::CS;
  float* Cs = smem;
  auto epilogue = [&](const TileCoord& tc) {
    const int t = opaque_tid();
    const int eli = t & 31, elh = (t >> 5) & 1;
    __syncthreads();   // every wave is done with the last chunk's operands
    if constexpr (POST != 0) {
      // ---- stage 1 output h = act(acc * scale + shift), written to LDS as split-f16 rows
      // [BN hi halves | BN lo halves] (row stride CS floats) = the A-operand image of stage 2
      _Float16* Hh = reinterpret_cast<_Float16*>(smem);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cl = wave_n * 32 + 8 * g + 4 * elh;
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cl + e < a.c_out) { sc[e] = a.scale[cl + e]; sh[e] = a.shift[cl + e]; }
#pragma unroll
        for (int wm = 0; wm < WTM; ++wm) {
          const int m = (wave_m * WTM + wm) * 32 + eli;
          half4 hi, lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[wm][0][4 * g + e] * sc[e] + sh[e];
            if (a.relu) v = fmaxf(v, 0.f);
            v = fminf(fmaxf(v, -65504.f), 65504.f);
            hi[e] = (_Float16)v;
            lo[e] = (_Float16)(v - (float)hi[e]);
          }
          *reinterpret_cast<half4*>(Hh + (size_t)m * CS * 2 + cl) = hi;
          *reinterpret_cast<half4*>(Hh + (size_t)m * CS * 2 + BN + cl) = lo;
        }
      }
      __syncthreads();
      // ---- stage 2: out2[pixel][n2] = sum_k h[pixel][k] * W2[n2][k], split-f16 x3
#pragma unroll
      for (int wm = 0; wm < WTM; ++wm)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[wm][0][r] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < BN / 16; ++s2) {
#pragma unroll
        for (int wm = 0; wm < WTM; ++wm) {
          const int m = (wave_m * WTM + wm) * 32 + eli;
          const half8 ah = *reinterpret_cast<const half8*>(&smem[m * CS + 8 * s2 + 4 * elh]);
          const half8 al = *reinterpret_cast<const half8*>(&smem[m * CS + BN / 2 + 8 * s2 + 4 * elh]);
          acc[wm][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2l[s2], ah, acc[wm][0], 0, 0, 0);
          acc[wm][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h[s2], al, acc[wm][0], 0, 0, 0);
          acc[wm][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h[s2], ah, acc[wm][0], 0, 0, 0);
        }
      }
      __syncthreads();   // h has been read by every wave
      // ---- stage 2 affine (+ReLU) -> fp32 rows [pixel][64] in LDS
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cl = wave_n * 32 + 8 * g + 4 * elh;
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (cl + e < a.c_out2) { sc[e] = a.scale2[cl + e]; sh[e] = a.shift2[cl + e]; }
#pragma unroll
        for (int wm = 0; wm < WTM; ++wm) {
          const int m = (wave_m * WTM + wm) * 32 + eli;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[wm][0][4 * g + e] * sc[e] + sh[e];
            if (a.relu2) v[e] = fmaxf(v[e], 0.f);
          }
          *reinterpret_cast<f32x4*>(&smem[m * CS + cl]) = v;
        }
      }
      __syncthreads();
      // ---- row-major stores; columns [0, split2) -> out (ldo), the rest -> out_b (ldo_b)
      const int nc4 = a.c_out2 >> 2, sp4 = a.split2 >> 2;
      const size_t img_px = (size_t)tc.img * a.h_out * a.w_out;
      for (int idx = t; idx < T::BM * nc4; idx += NT) {
        const int m = idx / nc4, c4 = idx % nc4;
        const int oy = tc.oy0 + m / TW, ox = tc.ox0 + m % TW;
        if (oy < a.h_out && ox < a.w_out) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(&smem[m * CS + 4 * c4]);
          const size_t px = img_px + (size_t)oy * a.w_out + ox;
          if (c4 < sp4) *reinterpret_cast<f32x4*>(a.out + px * a.ldo + 4 * c4) = v;
          else *reinterpret_cast<f32x4*>(a.out_b + px * a.ldo_b + 4 * (c4 - sp4)) = v;
        }
      }
      return;
    }
#pragma unroll
    for (int wn = 0; wn < WTN; ++wn) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cl = (wave_n * WTN + wn) * 32 + 8 * g + 4 * elh;   // channel within the tile
        const int co = tc.n0 + cl;
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (a.vec_out && co + 4 <= a.c_out) {
          sc = *reinterpret_cast<const f32x4*>(a.scale + co);
          sh = *reinterpret_cast<const f32x4*>(a.shift + co);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < a.c_out) { sc[e] = a.scale[co + e]; sh[e] = a.shift[co + e]; }
        }
#pragma unroll
        for (int wm = 0; wm < WTM; ++wm) {
          const int m = (wave_m * WTM + wm) * 32 + eli;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[wm][wn][4 * g + e] * sc[e] + sh[e];
            if (a.relu) v[e] = fmaxf(v[e], 0.f);
          }
          *reinterpret_cast<f32x4*>(&Cs[m * CS + cl]) = v;
        }
      }
    }
    __syncthreads();
    const int ncol = min(BN, a.c_out - tc.n0);             // valid channels of this tile
    float* obase = a.out + (size_t)tc.img * a.out_img + tc.n0;
    if (a.vec_out && (ncol & 3) == 0) {
      const int nc4 = ncol >> 2;
      for (int idx = t; idx < T::BM * nc4; idx += NT) {
        const int m = idx / nc4, c4 = idx % nc4;
        const int oy = tc.oy0 + m / TW, ox = tc.ox0 + m % TW;
        if (oy < a.h_out && ox < a.w_out) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[m * CS + 4 * c4]);
          if (!kNoStore || v[0] == 12345.678f)
            *reinterpret_cast<f32x4*>(obase + (size_t)oy * a.out_row + (size_t)ox * a.out_px + 4 * c4) = v;
        }
      }
    } else {
      for (int idx = t; idx < T::BM * ncol; idx += NT) {
        const int m = idx / ncol, c = idx % ncol;
        const int oy = tc.oy0 + m / TW, ox = tc.ox0 + m % TW;
        if (oy < a.h_out && ox < a.w_out) {
          const float v = Cs[m * CS + c];
          if (!kNoStore || v == 12345.678f) obase[(size_t)oy * a.out_row + (size_t)ox * a.out_px + c] = v;
        }
      }
    }
  };

  int item = blockIdx.x;
  if (item >= a.total_items) return;
  TileCoord cur = decode(item);
  setup_tile(cur);
  load_chunk(cur, 0);
  store_chunk();
  __syncthreads();

  while (true) {
#pragma unroll
    for (int wm = 0; wm < WTM; ++wm)
#pragma unroll
      for (int wn = 0; wn < WTN; ++wn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[wm][wn][r] = 0.f;

    bool has_next = false;
    TileCoord nxt = cur;
    for (int ch = 0; ch < a.nchunks; ++ch) {
      const bool last = ch + 1 == a.nchunks;
      // what to prefetch under this chunk's MFMAs: the tile's next chunk, or the
      // first chunk of the workgroup's next tile.  ONE load site: two would make
      // hipcc merge the staging registers with copies that wait for the loads
      // before the MFMA block instead of after it.
      bool stage = !last && !kNoStream;
      int stage_ch = ch + 1;
      if (last && item + G < a.total_items) {
        has_next = true;
        nxt = decode(item + G);
        setup_tile(nxt);
        stage = true;
        stage_ch = 0;
      }
      if (stage) load_chunk(nxt, stage_ch);   // nxt == cur until the tile's last chunk
      mfma_chunk();
      if (stage && !last) {
        __syncthreads();   // every wave is done reading this chunk from LDS
        store_chunk(true);
        __syncthreads();
      }
    }
    epilogue(cur);
    if (!has_next) break;
    __syncthreads();       // the staged output tile has been read back
    store_chunk();         // next tile's chunk 0, loaded under the last MFMA block
    __syncthreads();
    item += G;
    cur = nxt;
  }
}

// Persistent-workgroup policy: 0 = never, 1 = only the 256-pixel / fused tiles with 2-4 chunks,
// 2 = always (default).  History on MI355X, batch 4: with one step at a time and the first
// (channel-block-major) work-item order, policy 1 measured best (persistence +4-13 % on the short
// 32- and 64-channel full/half-resolution layers, -2-20 % on stride-2 and long-K launches).  With
// the XCD-aware order, interleaved A/B runs put "always" 1.5-2 % ahead of policy 1 (split-f16)
// and 3.4 % (exact fp32); "never" lands within 0.5 % of "always".
int g_persist = 2;

// ---------------------------------------------------------------------------
// tile menu and per-launch selection
// ---------------------------------------------------------------------------
enum CfgId {
  T3_256x32,   // 8x32 px, 32 ch
  T3_256x64,   // 8x32 px, 64 ch
  T3_128x64,   // 8x16 px, 64 ch
  T3_64x64,    // 8x8 px,  64 ch : small maps -> enough workgroups to fill 256 CUs
  T3S2_64x64,  // stride 2 (KC 8 in fp32, 16 in split-f16)
  T1_256x32, T1_256x64, T1_128x128, T1_64x64,
  CFG_COUNT
};

struct Cfg {
  CfgId id;
  int th, tw, bn;
  float bias[2];   // measured time per unit of tile area relative to 256x32, per math mode
};

// biases from profiles/r01_conv_tile_sweep.txt (tools/conv_tile_sweep.hip, batch-4
// layer shapes): in fp32 only the 64x64 tile pays (~10 %, one MFMA tile per wave);
// in split-f16 the MFMA phase is 5x shorter, so LDS reads per MFMA and weight
// staging per workgroup decide: the narrow 32-channel tile wins for deep layers.
const Cfg kCfgs[CFG_COUNT] = {
    {T3_256x32, 8, 32, 32, {1.00f, 1.00f}},   {T3_256x64, 8, 32, 64, {1.00f, 0.90f}},
    {T3_128x64, 8, 16, 64, {1.00f, 1.04f}},   {T3_64x64, 8, 8, 64, {1.10f, 1.45f}},
    {T3S2_64x64, 8, 8, 64, {1.00f, 1.00f}},   {T1_256x32, 8, 32, 32, {1.00f, 1.00f}},
    {T1_256x64, 8, 32, 64, {1.00f, 1.00f}},   {T1_128x128, 8, 16, 128, {1.00f, 1.00f}},
    {T1_64x64, 8, 8, 64, {1.00f, 1.00f}},
};

inline int out_dim(int in, int ksize, int stride) {
  const int pad = ksize / 2;
  return (in + 2 * pad - ksize) / stride + 1;
}

// MFMA-bound cost model: workgroups are dealt round-robin to the 256 CUs and a
// CU runs its workgroups' MFMAs on the same 4 SIMDs, so time ~ ceil(blocks/256)
// x (pixels x channels per tile) x the tile's measured bias; ties go to the
// earlier candidate.  Tiles wider than the padded channel count or than the map
// waste MFMAs and are charged for it by construction.
Cfg select_cfg(const dn_conv_desc& d) {
  const int ho = out_dim(d.h_in, d.ksize, d.stride), wo = out_dim(d.w_in, d.ksize, d.stride);
  const CfgId* cand;
  int ncand;
  // (a 16x32-pixel, 32-channel tile for the full-resolution layers was measured 3.4 % slower
  // per step than 8x32: fewer, longer workgroups hide less of the HBM latency)
  static const CfgId c3[] = {T3_256x32, T3_256x64, T3_128x64, T3_64x64};
  static const CfgId c3s2[] = {T3S2_64x64};
  static const CfgId c1[] = {T1_128x128, T1_256x64, T1_64x64, T1_256x32};
  if (d.ksize == 3 && d.stride == 2) { cand = c3s2; ncand = 1; }
  else if (d.ksize == 3) { cand = c3; ncand = 4; }
  else { cand = c1; ncand = 4; }
  Cfg best = kCfgs[cand[0]];
  double best_cost = 1e300;
  for (int k = 0; k < ncand; ++k) {
    const Cfg& c = kCfgs[cand[k]];
    const long tiles = (long)d.n_images * ((ho + c.th - 1) / c.th) * ((wo + c.tw - 1) / c.tw);
    const long blocks = tiles * ((d.c_out + c.bn - 1) / c.bn);
    const double rounds = (double)((blocks + 255) / 256);
    const double cost = rounds * c.th * c.tw * c.bn * c.bias[d.math == 1 ? 1 : 0];
    if (cost < best_cost * 0.999) { best_cost = cost; best = c; }
  }
  return best;
}

int validate(const dn_conv_desc* d) {
  DN_REQUIRE(d != nullptr, "conv: null descriptor");
  DN_REQUIRE(d->ksize == 1 || d->ksize == 3, "conv: ksize %d unsupported (1 or 3)", d->ksize);
  DN_REQUIRE(d->stride == 1 || (d->stride == 2 && d->ksize == 3),
             "conv: stride %d with ksize %d unsupported", d->stride, d->ksize);
  DN_REQUIRE(d->n_images > 0 && d->h_in > 0 && d->w_in > 0, "conv: empty input");
  DN_REQUIRE(d->c0 > 0 && d->c1 >= 0 && d->c_out > 0, "conv: bad channel counts");
  DN_REQUIRE(d->ld0 >= d->c0 && (d->c1 == 0 || d->ld1 >= d->c1) && d->ldo >= d->c_out,
             "conv: pixel strides smaller than channel counts");
  DN_REQUIRE(!d->up0 || (d->h_in % 2 == 0 && d->w_in % 2 == 0),
             "conv: x2-upsampled source needs even h_in/w_in");
  DN_REQUIRE(d->math == 0 || d->math == 1, "conv: math mode %d unknown (0 = fp32, 1 = split-f16)", d->math);
  DN_REQUIRE(d->c1 == 0 || d->c0 % kcp_of(d->ksize) == 0,
             "conv: concat needs c0 (%d) to be a multiple of %d", d->c0, kcp_of(d->ksize));
  return DN_OK;
}

inline int cout_pad_of(const dn_conv_desc& d) { return (d.c_out + 31) / 32 * 32; }
inline int nchunks_packed(const dn_conv_desc& d) {
  return (d.c0 + d.c1 + kcp_of(d.ksize) - 1) / kcp_of(d.ksize);
}

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wpk,
                                    int c_out, int c_in, int taps, int cout_pad, int kcp,
                                    long total) {
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int k = r % kcp; r /= kcp;
    const int co = r % cout_pad; r /= cout_pad;
    const int tap = r % taps;
    const int chp = r / taps;
    const int ci = chp * kcp + k;
    float v = 0.f;
    if (co < c_out && ci < c_in) v = w[((size_t)co * c_in + ci) * taps + tap];
    wpk[idx] = v;
  }
}

// split-f16 rows: [kcp hi halves | kcp lo halves] in the bytes of kcp floats
__global__ void pack_weights_split_kernel(const float* __restrict__ w, _Float16* __restrict__ wpk,
                                          int c_out, int c_in, int taps, int cout_pad, int kcp,
                                          long total) {
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int k = r % kcp; r /= kcp;
    const int co = r % cout_pad; r /= cout_pad;
    const int tap = r % taps;
    const int chp = r / taps;
    const int ci = chp * kcp + k;
    float v = 0.f;
    if (co < c_out && ci < c_in) v = w[((size_t)co * c_in + ci) * taps + tap];
    v = fminf(fmaxf(v, -65504.f), 65504.f);
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const long row = idx / kcp;
    wpk[row * 2 * kcp + k] = hi;
    wpk[row * 2 * kcp + kcp + k] = lo;
  }
}

// ---- many packs of this engine in ONE launch (dn_conv_pack_weights_multi; the split-planar engine's twin is in conv_sp.hip).
// A job = one call of pack_weights_kernel (split = 0) or pack_weights_split_kernel (split = 1) on the weight tensor a VIEW of
// `w` defines, times the power of two wmul (what the training engine multiplied in with a torch op before it packed):
//   mode 0: W[n][ci][t] = w[n][ci_first + ci][t]                   (the tensor itself, or a column cut of it)
//   mode 1: W[n][ci][t] = w[ci][ci_first + n][taps - 1 - t]        (dn_conv_dgrad_weights: the data gradient's conv)
struct NhwcPackJob {
  const float* w;
  void* out;
  int c_out, c_in, taps, cout_pad, kcp, mode, cin_total, ci_first, split, block_first, n_blocks, pad_;
  float wmul, padf_;
  long total;
};
static_assert(sizeof(NhwcPackJob) == 80, "dn_conv_pack_multi_table_bytes");

__global__ void __launch_bounds__(256) pack_weights_multi_kernel(const NhwcPackJob* __restrict__ jobs, int n_jobs) {
  __shared__ NhwcPackJob job;
  if (threadIdx.x == 0) {
    int lo = 0, hi = n_jobs - 1;                 // the last job whose first block is <= blockIdx.x
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].block_first <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    job = jobs[lo];
  }
  __syncthreads();
  const NhwcPackJob& j = job;
  const int vb = blockIdx.x - j.block_first;
  for (long idx = vb * (long)blockDim.x + threadIdx.x; idx < j.total; idx += (long)j.n_blocks * blockDim.x) {
    long r = idx;
    const int k = r % j.kcp; r /= j.kcp;
    const int co = r % j.cout_pad; r /= j.cout_pad;
    const int tap = r % j.taps;
    const int chp = r / j.taps;
    const int ci = chp * j.kcp + k;
    float v = 0.f;
    if (co < j.c_out && ci < j.c_in) {
      v = j.mode == 0 ? j.w[((size_t)co * j.cin_total + j.ci_first + ci) * j.taps + tap]
                      : j.w[((size_t)ci * j.cin_total + j.ci_first + co) * j.taps + (j.taps - 1 - tap)];
      v *= j.wmul;
    }
    if (j.split) {
      v = fminf(fmaxf(v, -65504.f), 65504.f);
      const _Float16 hi = (_Float16)v;
      const _Float16 lo = (_Float16)(v - (float)hi);
      const long row = idx / j.kcp;
      _Float16* o = static_cast<_Float16*>(j.out);
      o[row * 2 * j.kcp + k] = hi;
      o[row * 2 * j.kcp + j.kcp + k] = lo;
    } else {
      static_cast<float*>(j.out)[idx] = v;
    }
  }
}

__global__ void fold_bn_kernel(const float* bias, const float* gamma, const float* beta,
                               const float* mean, const float* var, float eps, int n,
                               float* scale, float* shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float b = bias ? bias[i] : 0.f;
  if (gamma) {
    const float sc = gamma[i] / sqrtf(var[i] + eps);
    scale[i] = sc;
    shift[i] = (b - mean[i]) * sc + beta[i];
  } else {
    scale[i] = 1.f;
    shift[i] = b;
  }
}

template <int KS, int STRIDE, int TH, int TW, int BN, int KC, int WAVES_M, int WAVES_N,
          int WTM, int WTN, int ABL = 0, int MATH = 0, int POST = 0>
int launch(ConvArgs& a, const dn_conv_desc& d, hipStream_t stream) {
  using T = ConvTile<KS, STRIDE, TH, TW, BN, KC, WAVES_M, WAVES_N, WTM, WTN>;
  auto kern = conv_mfma_kernel<KS, STRIDE, TH, TW, BN, KC, WAVES_M, WAVES_N, WTM, WTN, ABL, MATH, POST>;
  static_assert(POST == 0 || (size_t)T::BM * T::CS * sizeof(float) <= T::LDS_BYTES,
                "fused stage does not fit the tile's LDS");
  static_assert(T::LDS_BYTES <= 160 * 1024, "tile does not fit the 160 KiB LDS");
  // opt in to > 64 KiB of dynamic LDS once per instantiation (idempotent; a race
  // between two first callers only repeats the same attribute write)
  static dn::PerDeviceFlag lds_flag;
  bool& lds_ready = lds_flag.here();
  static int occupancy = 1;
  if (!lds_ready) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)T::LDS_BYTES);
    if (e != hipSuccess)
      return dn::fail(DN_ERR_LAUNCH, "conv: hipFuncSetAttribute(%zu B LDS): %s", T::LDS_BYTES,
                      hipGetErrorString(e));
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, T::NT, T::LDS_BYTES) == hipSuccess &&
        occ >= 1)
      occupancy = occ > 8 ? 8 : occ;
    lds_ready = true;
  }
  a.nchunks = (d.c0 + d.c1 + KC - 1) / KC;
  a.tiles_x = (a.w_out + TW - 1) / TW;
  a.tiles_y = (a.h_out + TH - 1) / TH;
  const long total = (long)a.n_images * a.tiles_y * a.tiles_x * ((d.c_out + BN - 1) / BN);
  DN_REQUIRE(total < (1L << 31), "conv: too many tiles (%ld)", total);
  a.total_items = (int)total;
  // DN_CONV_XCD=0 restores channel-block-major order, 1 interleaves pixel tiles over the
  // XCDs.  Measured (batch 4): conv FETCH per step 3.54 GB (0) -> 2.79 GB (1) -> 2.49 GB (2);
  // step time within 1 % of each other -- the re-reads came out of the 256 MB MALL.
  static const int xcd_env = [] {
    const char* e = getenv("DN_CONV_XCD");
    return e ? atoi(e) : 2;
  }();
  a.xcd_order = xcd_env;
  // persistent grid: as many workgroups as are resident (no inter-workgroup sync
  // depends on the count; an over-estimate only queues the surplus)
  const long resident = (long)occupancy * kNumCUs;
  static const int persist_env = [] {   // experiments: DN_CONV_PERSIST=0|1|2 overrides the policy
    const char* e = getenv("DN_CONV_PERSIST");
    return e ? atoi(e) : -1;
  }();
  const int pmode = persist_env >= 0 ? persist_env : g_persist;
  // fused-1x1 launches: short tiles with a long epilogue and per-workgroup weight registers
  const bool persist = pmode == 2 || (pmode == 1 && (POST != 0 || T::BM >= 256) && a.nchunks >= 2 &&
                                      a.nchunks <= 4);
  dim3 grid((unsigned)((persist && total > resident) ? resident : total));
  hipLaunchKernelGGL(kern, grid, dim3(T::NT), T::LDS_BYTES, stream, a);
  return dn::check_launch("conv_mfma_kernel");
}

}  // namespace

extern "C" size_t dn_conv_packed_weight_floats(const dn_conv_desc* d) {
  if (validate(d) != DN_OK) return 0;
  return (size_t)nchunks_packed(*d) * d->ksize * d->ksize * cout_pad_of(*d) * kcp_of(d->ksize);
}

extern "C" int dn_conv_pack_weights(const dn_conv_desc* d, const float* weight_oihw,
                                    float* packed, void* stream) {
  if (int rc = validate(d)) return rc;
  DN_REQUIRE(weight_oihw && packed, "conv pack: null pointer");
  const long total = (long)dn_conv_packed_weight_floats(d);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (d->math == 1)
    hipLaunchKernelGGL(pack_weights_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       weight_oihw, reinterpret_cast<_Float16*>(packed), d->c_out, d->c0 + d->c1,
                       d->ksize * d->ksize, cout_pad_of(*d), kcp_of(d->ksize), total);
  else
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       weight_oihw, packed, d->c_out, d->c0 + d->c1, d->ksize * d->ksize,
                       cout_pad_of(*d), kcp_of(d->ksize), total);
  return dn::check_launch("pack_weights_kernel");
}

extern "C" size_t dn_conv_pack_multi_table_bytes(int n_jobs) { return n_jobs > 0 ? sizeof(NhwcPackJob) * (size_t)n_jobs : 0; }

extern "C" int dn_conv_pack_multi_prepare(const dn_pack_job* jobs, int n_jobs, void* table_host, int* total_blocks) {
  DN_REQUIRE(jobs && table_host && total_blocks && n_jobs > 0, "conv pack multi: null pointer / no jobs");
  NhwcPackJob* t = static_cast<NhwcPackJob*>(table_host);
  int first = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const dn_pack_job& q = jobs[i];
    const dn_conv_desc* d = &q.desc;
    if (int rc = validate(d)) return rc;
    DN_REQUIRE(q.weight && q.packed, "conv pack multi: job %d: null pointer", i);
    DN_REQUIRE(q.mode == 0 || q.mode == 1, "conv pack multi: job %d: mode %d (0 or 1)", i, q.mode);
    const int c_in = d->c0 + d->c1;
    DN_REQUIRE(q.ci_first >= 0 && q.ci_first + (q.mode == 0 ? c_in : d->c_out) <= q.cin_total,
               "conv pack multi: job %d: columns %d + %d of %d", i, q.ci_first, q.mode == 0 ? c_in : d->c_out, q.cin_total);
    NhwcPackJob& j = t[i];
    j.w = q.weight;
    j.out = q.packed;
    j.c_out = d->c_out; j.c_in = c_in; j.taps = d->ksize * d->ksize; j.cout_pad = cout_pad_of(*d); j.kcp = kcp_of(d->ksize);
    j.mode = q.mode; j.cin_total = q.cin_total; j.ci_first = q.ci_first; j.split = d->math == 1;
    j.wmul = q.wmul; j.pad_ = 0; j.padf_ = 0.f;
    j.total = (long)dn_conv_packed_weight_floats(d);
    j.n_blocks = (int)((j.total + 255) / 256 < 4096 ? (j.total + 255) / 256 : 4096);      // the single launch's grid
    j.block_first = first;
    first += j.n_blocks;
  }
  *total_blocks = first;
  return DN_OK;
}

extern "C" int dn_conv_pack_weights_multi(const void* table_device, int n_jobs, int total_blocks, void* stream) {
  DN_REQUIRE(table_device && n_jobs > 0 && total_blocks > 0, "conv pack multi: bad arguments");
  hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const NhwcPackJob*>(table_device), n_jobs);
  return dn::check_launch("pack_weights_multi_kernel");
}

extern "C" int dn_fold_bn(const float* bias, const float* gamma, const float* beta,
                          const float* mean, const float* var, float eps, int channels,
                          float* scale, float* shift, void* stream) {
  DN_REQUIRE(channels > 0 && scale && shift, "fold_bn: bad arguments");
  DN_REQUIRE(!gamma || (beta && mean && var), "fold_bn: gamma without beta/mean/var");
  hipLaunchKernelGGL(fold_bn_kernel, dim3((channels + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, bias, gamma, beta, mean, var, eps, channels, scale,
                     shift);
  return dn::check_launch("fold_bn_kernel");
}

namespace {

__global__ void pack_post1x1_kernel(const float* __restrict__ w2, _Float16* __restrict__ out,
                                    int c_out2, int c_in2) {
  // [64 rows][64 hi halves | 64 lo halves]; rows >= c_out2 / columns >= c_in2 are zero
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 64 * 64) return;
  const int n = idx / 64, k = idx % 64;
  float v = (n < c_out2 && k < c_in2) ? w2[(size_t)n * c_in2 + k] : 0.f;
  v = fminf(fmaxf(v, -65504.f), 65504.f);
  const _Float16 hi = (_Float16)v;
  out[n * 128 + k] = hi;
  out[n * 128 + 64 + k] = (_Float16)(v - (float)hi);
}

int fill_args(const dn_conv_desc* d, const float* src0, const float* src1, const float* packed,
              const float* scale, const float* shift, float* out, ConvArgs& a);

}  // namespace

extern "C" size_t dn_post1x1_packed_floats(void) { return 64 * 64; }

extern "C" int dn_post1x1_pack_weights(const float* w2, int c_out2, int c_in2, float* packed,
                                       void* stream) {
  DN_REQUIRE(w2 && packed, "post1x1 pack: null pointer");
  DN_REQUIRE(c_out2 > 0 && c_out2 <= 64 && c_in2 > 0 && c_in2 <= 64,
             "post1x1 pack: c_out2 %d / c_in2 %d must be in 1..64", c_out2, c_in2);
  hipLaunchKernelGGL(pack_post1x1_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, w2,
                     reinterpret_cast<_Float16*>(packed), c_out2, c_in2);
  return dn::check_launch("pack_post1x1_kernel");
}

extern "C" int dn_conv2d_post1x1(const dn_conv_desc* d, const dn_post1x1_desc* p,
                                 const float* src0, const float* src1, const float* packed,
                                 const float* scale, const float* shift, const float* packed2,
                                 const float* scale2, const float* shift2, float* out_a,
                                 float* out_b, void* stream) {
  if (int rc = validate(d)) return rc;
  DN_REQUIRE(p && src0 && packed && scale && shift && packed2 && scale2 && shift2 && out_a,
             "conv+1x1: null pointer");
  DN_REQUIRE(d->math == 1 && d->ksize == 3 && d->stride == 1 && d->c_out == 64,
             "conv+1x1: needs the split-f16 3x3 stride-1 path with 64 output channels");
  DN_REQUIRE(p->c_out2 > 0 && p->c_out2 <= 64 && p->c_out2 % 4 == 0 && p->split % 4 == 0 &&
                 p->split > 0 && p->split <= p->c_out2,
             "conv+1x1: c_out2 %d / split %d must be multiples of 4, split in (0, c_out2]",
             p->c_out2, p->split);
  DN_REQUIRE(p->ldo_a >= p->split && p->ldo_a % 4 == 0, "conv+1x1: bad ldo_a");
  DN_REQUIRE(p->split == p->c_out2 || (out_b && p->ldo_b >= p->c_out2 - p->split && p->ldo_b % 4 == 0),
             "conv+1x1: second output missing or too narrow");
  auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  DN_REQUIRE(aligned16(out_a) && aligned16(packed2) && (!out_b || aligned16(out_b)),
             "conv+1x1: outputs / packed2 must be 16-byte aligned");
  ConvArgs a;
  if (int rc = fill_args(d, src0, src1, packed, scale, shift, out_a, a)) return rc;
  a.ldo = p->ldo_a;
  a.w2 = packed2; a.scale2 = scale2; a.shift2 = shift2; a.out_b = out_b;
  a.c_out2 = p->c_out2; a.relu2 = p->relu2; a.split2 = p->split; a.ldo_b = p->ldo_b;
  // 8x32-pixel workgroups (4 MFMA tiles per wave): +2.9 % per step over 8x16 (weights staged and
  // barriers paid half as often per pixel); DN_POST_TILE=128 selects the smaller tile
  static const int tile_env = [] {
    const char* e = getenv("DN_POST_TILE");
    return e ? atoi(e) : 256;
  }();
  if (tile_env == 256) return launch<3, 1, 8, 32, 64, 16, 2, 2, 4, 1, 0, 1, 1>(a, *d, (hipStream_t)stream);
  return launch<3, 1, 8, 16, 64, 16, 2, 2, 2, 1, 0, 1, 1>(a, *d, (hipStream_t)stream);
}

namespace {

// common argument set-up of dn_conv2d / dn_conv2d_post1x1
int fill_args(const dn_conv_desc* d, const float* src0, const float* src1, const float* packed,
              const float* scale, const float* shift, float* out, ConvArgs& a) {
  a.src0 = src0; a.src1 = src1; a.wpk = packed; a.scale = scale; a.shift = shift; a.out = out;
  a.n_images = d->n_images; a.h_in = d->h_in; a.w_in = d->w_in;
  a.h_out = out_dim(d->h_in, d->ksize, d->stride);
  a.w_out = out_dim(d->w_in, d->ksize, d->stride);
  a.c0 = d->c0; a.c1 = d->c1; a.up0 = d->up0; a.c_out = d->c_out; a.relu = d->relu;
  a.ld0 = d->ld0; a.ld1 = d->ld1; a.ldo = d->ldo;
  a.cout_pad = cout_pad_of(*d);
  a.wpk_bytes = (int)(dn_conv_packed_weight_floats(d) * sizeof(float));
  a.w2 = nullptr; a.scale2 = nullptr; a.shift2 = nullptr; a.out_b = nullptr;
  a.c_out2 = 0; a.relu2 = 0; a.split2 = 0; a.ldo_b = 0;
  a.tap_mask = 0x1ff;
  a.out_px = d->ldo; a.out_row = a.w_out * d->ldo; a.out_img = (long)a.h_out * a.w_out * d->ldo;
  auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  a.vec0 = (d->c0 % 4 == 0 && d->ld0 % 4 == 0 && aligned16(src0)) ? 1 : 0;
  a.vec1 = (d->c1 > 0 && d->c1 % 4 == 0 && d->ld1 % 4 == 0 && aligned16(src1)) ? 1 : 0;
  DN_REQUIRE(aligned16(packed), "conv: packed weights must be 16-byte aligned");
  a.vec_out = (d->ldo % 4 == 0 && aligned16(out) && aligned16(scale) && aligned16(shift)) ? 1 : 0;
  // buffer descriptors address one image with 32-bit byte offsets
  const size_t hs0 = d->up0 ? d->h_in / 2 : d->h_in, ws0 = d->up0 ? d->w_in / 2 : d->w_in;
  DN_REQUIRE(hs0 * ws0 * d->ld0 * 4 < (1ull << 31) &&
                 (size_t)d->h_in * d->w_in * (d->c1 ? d->ld1 : 1) * 4 < (1ull << 31),
             "conv: one image must stay below 2 GiB");
  return DN_OK;
}

}  // namespace

namespace {
int conv2d_impl(const dn_conv_desc* d, const float* src0, const float* src1, const float* packed, const float* scale,
                const float* shift, float* out, int tap_mask, long out_img, int out_row, int out_px, void* stream);
}

extern "C" int dn_conv2d(const dn_conv_desc* d, const float* src0, const float* src1,
                         const float* packed, const float* scale, const float* shift,
                         float* out, void* stream) {
  return conv2d_impl(d, src0, src1, packed, scale, shift, out, 0x1ff, 0, 0, 0, stream);
}

extern "C" int dn_conv2d_taps(const dn_conv_desc* d, const float* src0, const float* src1, const float* packed,
                              const float* scale, const float* shift, float* out, int tap_mask, long out_img_stride,
                              int out_row_stride, int out_px_stride, void* stream) {
  DN_REQUIRE(d && d->ksize == 3 && tap_mask > 0 && tap_mask <= 0x1ff, "conv (taps): a 3x3 layer and a non-empty 9-bit mask");
  DN_REQUIRE(out_px_stride >= d->c_out && out_row_stride > 0 && out_img_stride > 0,
             "conv (taps): output strides (floats) must be positive, pixel stride >= c_out");
  return conv2d_impl(d, src0, src1, packed, scale, shift, out, tap_mask, out_img_stride, out_row_stride, out_px_stride, stream);
}

namespace {
int conv2d_impl(const dn_conv_desc* d, const float* src0, const float* src1, const float* packed, const float* scale,
                const float* shift, float* out, int tap_mask, long out_img, int out_row, int out_px, void* stream) {
  if (int rc = validate(d)) return rc;
  DN_REQUIRE(src0 && packed && scale && shift && out, "conv: null pointer");
  DN_REQUIRE(d->c1 == 0 || src1, "conv: c1 > 0 but src1 is null");
  const Cfg c = select_cfg(*d);
  ConvArgs a;
  if (int rc = fill_args(d, src0, src1, packed, scale, shift, out, a)) return rc;
  a.tap_mask = tap_mask;
  if (out_img > 0) {
    a.out_img = out_img; a.out_row = out_row; a.out_px = out_px;
    a.vec_out = a.vec_out && out_px % 4 == 0 && out_row % 4 == 0 && out_img % 4 == 0;
  }
  hipStream_t s = (hipStream_t)stream;
#define DN_CONV_CASE(ID, ...)                                                         \
  case ID:                                                                            \
    return d->math == 1 ? launch<__VA_ARGS__, 0, 1>(a, *d, s) : launch<__VA_ARGS__, 0, 0>(a, *d, s);
  switch (c.id) {
    //                        KS S  TH TW  BN  KC WM WN WTM WTN
    DN_CONV_CASE(T3_256x32,   3, 1, 8, 32, 32, 16, 4, 1, 2, 1)
    DN_CONV_CASE(T3_256x64,   3, 1, 8, 32, 64, 16, 4, 1, 2, 2)
    DN_CONV_CASE(T3_128x64,   3, 1, 8, 16, 64, 16, 2, 2, 2, 1)
    DN_CONV_CASE(T3_64x64,    3, 1, 8, 8, 64, 16, 2, 2, 1, 1)
    case T3S2_64x64:   // 8-channel chunks keep 3 workgroups/CU in fp32; split-f16 wants 16 k per MFMA
      return d->math == 1 ? launch<3, 2, 8, 8, 64, 16, 2, 2, 1, 1, 0, 1>(a, *d, s)
                          : launch<3, 2, 8, 8, 64, 8, 2, 2, 1, 1, 0, 0>(a, *d, s);
    DN_CONV_CASE(T1_256x32,   1, 1, 8, 32, 32, 32, 4, 1, 2, 1)
    DN_CONV_CASE(T1_256x64,   1, 1, 8, 32, 64, 32, 4, 1, 2, 2)
    DN_CONV_CASE(T1_128x128,  1, 1, 8, 16, 128, 32, 2, 2, 2, 2)
    DN_CONV_CASE(T1_64x64,    1, 1, 8, 8, 64, 32, 2, 2, 1, 1)
    default: break;
  }
#undef DN_CONV_CASE
  return dn::fail(DN_ERR_UNSUPPORTED, "conv: no tile configuration");
}
}  // namespace
