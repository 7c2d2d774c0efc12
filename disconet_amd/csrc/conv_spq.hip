// Quad-merged 3x3 convolution over cat([nearest-upsample x2 (src0), src1]) -- the decoder's *_1 layers
// (upstream:coperception/models/det/backbone/Backbone.py :: decode: cat(interpolate(x_prev, x2), skip) ->
// conv -> BN -> ReLU) -- on the split-planar engine of conv_sp.hip (same SP tensors, same
// split-f16 x3 arithmetic, same LDS-DMA staging).
//
// Why a kernel of its own.  Over a nearest-upsampled source the 3x3 taps of an output pixel read only 2 x 2
// distinct low-resolution pixels; WHICH taps coincide depends on the pixel's parity class (py, px) =
// (oy & 1, ox & 1):
//     rows:  py = 0: {dy 0} {dy 1, dy 2}      py = 1: {dy 0, dy 1} {dy 2}        (columns alike with px)
// Summing the coinciding weights at pack time turns the 9 taps of the upsampled source's chunks into 4 merged
// taps per class: 4/9 of the MACs on c0 / (c0 + c1) of K (conv5_1..conv7_1: 2/3 of K -> 0.63 of the plain MAC
// count; the row-only merge of conv_sp.hip's UPM form reaches 0.78).  The price is that an MFMA pixel tile
// must be class-pure, so
//   * a workgroup owns an 8 x 32 output tile, its four waves ARE the four classes: wave w = py + 2 px computes
//     the 4 x 16 pixels {(py + 2 i, px + 2 k)} as two 32-pixel MFMA tiles, for BN = 32 or 64 channels;
//   * both sources are staged as the same (8 + 2) x (32 + 2) patch with DE-INTERLEAVED columns (even columns,
//     then odd ones: conv_sp.hip's stride-2 layout), so that the 16 lanes of a ds_read_b128 group -- one row,
//     16 columns of one parity -- read 16 consecutive pieces: conflict-free for every tap of every class.  The
//     upsample is an addressing mode of the patch DMA (sy = iy >> 1, sx = ix >> 1), as in conv_sp.hip;
//   * a step of the K loop stages, for a chunk of source 0, ONE merged tap x 4 classes of weights (BN = 64;
//     two taps at BN = 32) and every wave reads its own class block; for a chunk of source 1 three plain taps
//     shared by all waves.  The per-lane LDS bases absorb the class (registers, not immediates), the tap
//     offsets stay compile-time.
// Pipeline (double-buffered patch per chunk, double-buffered weights per step, raw s_barrier + counted
// vmcnt waits, next tile's first operands under the current tile's last step), persistent workgroups, XCD-aware
// item order and the register epilogue are those of conv_sp_kernel.
#ifndef DN_MFMA_PRIO
#define DN_MFMA_PRIO 0
#endif
#ifndef DN_MMA_GRAY
#define DN_MMA_GRAY 1      // see conv_sp.hip
#endif
#include "dn_internal.h"
#include "sp_layout.h"
#include "sp_device.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int kCUs = 256;

// patch geometry: PH x PW pixels, columns de-interleaved (QHALF even columns, then QHALF odd ones)
constexpr int QTH = 8, QTW = 32, QPH = QTH + 2, QPW = QTW + 2, QHALF = QPW / 2, QPITCH = QPW, QNPIX = QPH * QPITCH;
constexpr int QNW = 4, QNT = QNW * 64;
constexpr int QA_PIECES = 4 * QNPIX;                       // 4 quarters
constexpr int QA_INSTR = (QA_PIECES + 63) / 64;            // 22 DMA instructions per patch stage
constexpr int QA_IT = (QA_INSTR + QNW - 1) / QNW;          // rounds; the last one is ragged (waves 0, 1 only)
constexpr int QA_STAGE = QA_INSTR * 1024;
static_assert(QA_INSTR == 22 && QA_IT == 6, "patch stage: 22 instructions, waves 0 / 1 issue 6, waves 2 / 3 issue 5");

struct SpqArgs {
  const unsigned char* src0;
  const unsigned char* src1;
  const unsigned char* wpk;
  const float* scale;
  const float* shift;
  unsigned char* out;          // SP tensor; nullptr: the fp32 rows below are the only output
  float* out_f32;              // optional second output: the same values as fp32 NHWC rows (pixel stride ldo_f32 floats) --
  int ldo_f32;                 // the training step's forward (z = conv + bias, read by the BatchNorm statistics)
  int n_images, h, w;          // conv input = output size (src0 is stored at h / 2 x w / 2)
  int c0g, c1g;                // 16-channel chunks from src0 / src1
  int c_out, cog, relu;
  int tiles_x, tiles_y, total_items;
  int cout_pad, wpk_bytes;
  int n_cb;                    // channel blocks; reciprocals of the work-item decode's divisors (sp_device.h :: fdivmod)
  float rcp_ncb, rcp_tx, rcp_ty;
  KSlices ks;                  // K slices (KSL kernels; sp_device.h :: KSlices)
};

struct QTile {
  int img, oy0, ox0, n0;
};

// DEEP: one step per 16-channel chunk (all 4 merged taps x 4 classes of a source-0 chunk, all 9 taps of a source-1
// chunk).  A launch of fewer items than CUs (one rank's 4-image share of the agent-sharded step) runs at the latency
// of a step -- one workgroup per CU, nothing to switch to while its weight DMA is in flight -- so the step count is
// what it pays for: 48 steps instead of 112 on conv5_1.  One workgroup per CU (111 KB of LDS).
template <int BN, int DEEP = 0>
struct SpqTile {
  static constexpr int WTN = BN / 32;
  static constexpr int TGQ = DEEP ? 4 : (BN == 64 ? 1 : 2);   // merged taps of source 0 per step
  static constexpr int NS0 = 4 / TGQ;                  // steps per chunk of source 0
  static constexpr int TG1 = DEEP ? 9 : 3;             // taps of source 1 per step
  static constexpr int NS1 = 9 / TG1;
  static constexpr int BLK = 4 * BN * 16;              // bytes of one weight block [4 quarters][BN] in LDS
  static constexpr int B_PIECES0 = TGQ * 4 * 4 * BN;   // step of source 0: TGQ taps x 4 classes
  static constexpr int B_PIECES1 = TG1 * 4 * BN;       // step of source 1: TG1 taps
  static constexpr int B_IT0 = (B_PIECES0 + QNT - 1) / QNT;
  static constexpr int B_IT1 = (B_PIECES1 + QNT - 1) / QNT;
  static_assert(B_IT1 <= B_IT0, "the source-0 step is the larger one");
  static constexpr int QB_STAGE = B_IT0 * QNT * 16;    // one weight stage: 16 KB (16 blocks of BN = 32 or 4 of BN = 64); DEEP 32 KB
  static constexpr int AFF_BYTES = QNW * 2 * 64 * 4;   // per-wave copy of the epilogue affine (scale, shift) of a channel block
  static constexpr int LDS_BYTES = 2 * QA_STAGE + 2 * QB_STAGE + AFF_BYTES;
  static_assert((DEEP ? 1 : 2) * LDS_BYTES <= 160 * 1024, "two workgroups per CU (DEEP: one)");
};

// ABL (tools/sp_conv_check only, results are garbage): 1 = no weight DMA, 2 = no patch DMA, 3 = neither
// KSL: K-sliced launch (sp_device.h :: KSlices): whole tiles fold their slices in a second accumulator set, split
// tiles hand theirs through `ks.partial` to the fix-up pass.  BN = 32 only (at BN = 64 the second set does not fit).
template <int BN, int DEEP, int ABL = 0, int KSL = 0>
__global__ void __launch_bounds__(QNT, 2) conv_spq_kernel(const SpqArgs a) {
  static_assert(!KSL || BN == 32, "K slices: the 32-channel tile");
  using T = SpqTile<BN, DEEP>;
  constexpr int WTN = T::WTN, TGQ = T::TGQ, NS0 = T::NS0, NS1 = T::NS1, BLK = T::BLK, QB_STAGE = T::QB_STAGE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int OFF_B = 2 * QA_STAGE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int py = wave & 1, px = wave >> 1;                 // this wave's parity class

  // ---- work items (channel block, image, tile_y, tile_x), XCD-aware order as in conv_sp_kernel
  const int G = gridDim.x;
  const int spatial_items = a.n_images * a.tiles_y * a.tiles_x;
  const int n_cb = a.n_cb;
  const int sp_full = spatial_items & ~7;
  auto decode = [&](int item) {
    QTile tc;
    int cb, spi;
    if (item < sp_full * n_cb) {
      const int j = item >> 3;
      const int jq = fdivmod(j, n_cb, a.rcp_ncb, cb);
      spi = (item & 7) * (sp_full >> 3) + jq;
    } else {
      const int r = item - sp_full * n_cb, rem = spatial_items - sp_full;   // the last < 8 spatial items: rare
      cb = r / rem;
      spi = sp_full + r % rem;
    }
    tc.n0 = cb * BN;
    int tx, ty;
    spi = fdivmod(spi, a.tiles_x, a.rcp_tx, tx);
    tc.img = fdivmod(spi, a.tiles_y, a.rcp_ty, ty);
    tc.ox0 = tx * QTW;
    tc.oy0 = ty * QTH;
    return tc;
  };

  // sl0 .. sl1: the slices the item computes (all of them: a whole tile, folded in registers; one: a split tile's
  // slice `sl0`, partial index j = v - n_whole = tile * kslices + slice)
  struct Work {
    QTile tc;
    int sl0, sl1, j;
  };
  const int nchunks = a.c0g + a.c1g;
  auto decode_work = [&](int v) {
    Work w;
    if (KSL == 0 || v < a.ks.n_whole) {
      w.tc = decode(v);
      w.sl0 = 0; w.sl1 = KSL ? a.ks.count : 1; w.j = -1;
    } else {
      w.j = v - a.ks.n_whole;
      w.sl0 = w.j & ((1 << a.ks.log2) - 1);
      w.sl1 = w.sl0 + 1;
      w.tc = decode(a.ks.n_whole + (w.j >> a.ks.log2));
    }
    return w;
  };
  auto first_group = [&](const Work& w) { return KSL ? a.ks.bound(w.sl0) : 0; };

  // ---- this lane's pixels and LDS read bases (bytes).  MFMA tile wm of the class: tile rows
  // r = py + 4 wm + 2 rsel, columns c = px + 2 k; one 16-lane ds_read_b128 group = one row, k = 0..15.
  const int rsel = sp::in_g2(li) ? 1 : 0, kcol = sp::rank16(li);
  int prow[2], pcol;
  pcol = px + 2 * kcol;
  auto colpos = [](int e) { return (e & 1) * QHALF + (e >> 1); };   // row position of patch column 2 k + e, minus k
  int a_b1[2][3];        // source 1, tap (dy, dx): + dy * QPITCH * 16
  int a_b0[2][2][2];     // source 0, merged tap (a, b): [wm][b][a]
#pragma unroll
  for (int wm = 0; wm < 2; ++wm) {
    prow[wm] = py + 4 * wm + 2 * rsel;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) a_b1[wm][dx] = (lh * QNPIX + prow[wm] * QPITCH + colpos(px + dx) + kcol) * 16;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int am = 0; am < 2; ++am)   // merged tap (am, b) reads patch row r + am + (py & am), column c + b + (px & b)
        a_b0[wm][b][am] = (lh * QNPIX + (prow[wm] + am + (py & am)) * QPITCH + colpos(px + b + (px & b)) + kcol) * 16;
  }
  int b_off[WTN], b_offq[WTN];   // weight fragment of channel tile wn: block-relative, and + this wave's class block
#pragma unroll
  for (int wn = 0; wn < WTN; ++wn) {
    b_off[wn] = (lh * BN + wn * 32 + li) * 16;
    b_offq[wn] = b_off[wn] + wave * BLK;
  }

  f32x16 acc[2][WTN];
  f32x16 tot[KSL ? 2 : 1][KSL ? WTN : 1];   // K slices: the sum, in slice order, of the slices' accumulation chains
  float amax = 0.f;
  bool nan_seen = false;

  // ---- DMA state
  constexpr unsigned OOB = 0xFFFFFFFFu;
  const int hs0 = a.h >> 1, ws0 = a.w >> 1;
  const unsigned plane0 = (unsigned)(hs0 * ws0) * 16u, plane1 = (unsigned)(a.h * a.w) * 16u;
  const size_t img0_bytes = (size_t)a.c0g * 4 * plane0, img1_bytes = (size_t)a.c1g * 4 * plane1;
  auto rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.src0), 0, 0, 0x00020000);
  auto rsrc1 = rsrc0;
  const auto rsrcw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.wpk), 0, a.wpk_bytes, 0x00020000);
  unsigned voff_a[QA_IT], voff_b[T::B_IT0];

  auto opaque = [](int v) {
    asm volatile("" : "+v"(v));
    return v;
  };
  // per-lane source offsets of the patch pieces this lane moves: LDS piece (it * NW + wave) * 64 + lane.  The piece's
  // patch pixel (r, cc) and quarter plane are tile-independent: decoded once per launch into one packed register
  // (r | cc << 8 | cq << 16 | valid << 31), as in conv_sp_kernel; a tile adds its origin and tests the image bounds.
  unsigned piece_map[QA_IT];
#pragma unroll
  for (int it = 0; it < QA_IT; ++it) {
    const int piece = (it * QNW + wave) * 64 + lane;
    const int cq = piece / QNPIX, pp = piece % QNPIX;
    const int r = pp / QPITCH, pos = pp % QPITCH;
    const int cc = pos < QHALF ? 2 * pos : 2 * (pos - QHALF) + 1;
    piece_map[it] = (unsigned)r | ((unsigned)cc << 8) | ((unsigned)cq << 16) | (piece < QA_PIECES ? 0x80000000u : 0u);
  }
  auto setup_voff_a = [&](const QTile& tc, bool from1) {
    const int iy0 = tc.oy0 - 1, ix0 = tc.ox0 - 1;
    const unsigned plane = from1 ? plane1 : plane0;
    const int ws = from1 ? a.w : ws0;
#pragma unroll
    for (int it = 0; it < QA_IT; ++it) {
      unsigned pm = piece_map[it];
      asm volatile("" : "+v"(pm));
      const int r = pm & 0xff, cc = (pm >> 8) & 0xff, cq = (pm >> 16) & 0xff;
      const int iy = iy0 + r, ix = ix0 + cc;
      const bool ok = (int)pm < 0 && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w;
      const int sy = from1 ? iy : (iy >> 1), sx = from1 ? ix : (ix >> 1);
      voff_a[it] = ok ? (unsigned)cq * plane + (unsigned)(sy * ws + sx) * 16u : OOB;
    }
  };
  // weight pieces: LDS piece -> (block of the step, quarter, channel); the same map serves both step kinds
  int voffb_n0 = -1;
  auto setup_voff_b = [&](const QTile& tc) {
    if (tc.n0 == voffb_n0) return;
    voffb_n0 = tc.n0;
    const int t = opaque(tid);
#pragma unroll
    for (int it = 0; it < T::B_IT0; ++it) {
      const int piece = (it * QNW + (t >> 6)) * 64 + (t & 63);
      const int blk = piece / (4 * BN), q = (piece / BN) % 4, nn = piece % BN;
      voff_b[it] = (unsigned)(((blk * 4 + q) * a.cout_pad + tc.n0 + nn) * 16);
    }
  };
  auto setup_rsrc = [&](const QTile& tc) {
    rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.src0 + tc.img * img0_bytes), 0,
                                              (int)img0_bytes, 0x00020000);
    rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.c1g ? a.src1 + tc.img * img1_bytes : a.src0),
                                              0, a.c1g ? (int)img1_bytes : 0, 0x00020000);
  };
  // patch of chunk g -> stage sa
  auto issue_a = [&](int g, int sa) {
    if constexpr ((ABL & 2) != 0) return;
    const bool from1 = g >= a.c0g;
    const int soff = from1 ? (g - a.c0g) * 4 * (int)plane1 : g * 4 * (int)plane0;
    unsigned char* base = smem + sa * QA_STAGE + wave * 1024;
#pragma unroll
    for (int it = 0; it < QA_IT; ++it) {
      if (it == QA_IT - 1 && wave >= QA_INSTR - (QA_IT - 1) * QNW) break;   // ragged last round
      if (from1)
        dma16(rsrc1, base + it * QNW * 1024, voff_a[it], soff);
      else
        dma16(rsrc0, base + it * QNW * 1024, voff_a[it], soff);
    }
  };
  // weights of step st of chunk g -> stage sb.  Packed image: chunks of source 0 carry 16 blocks
  // [merged tap t = 2 a + b][class], the others 9 blocks [dy][dx]; a block = [4 quarters][cout_pad] pieces.
  auto issue_b = [&](int g, int st, int sb) {
    if constexpr ((ABL & 1) != 0) return;
    const bool from1 = g >= a.c0g;
    const int blk = from1 ? a.c0g * 16 + (g - a.c0g) * 9 + st * T::TG1 : g * 16 + st * (TGQ * 4);
    const int soff = blk * 4 * a.cout_pad * 16;
    unsigned char* base = smem + OFF_B + sb * QB_STAGE + wave * 1024;
#pragma unroll
    for (int it = 0; it < T::B_IT0; ++it)
      if (!from1 || it < T::B_IT1) dma16(rsrcw, base + it * QNW * 1024, voff_b[it], soff);
  };

  // ---- one sub-step = one tap: 2 x WTN accumulator tiles x 3 MFMAs
  struct Frags {
    half8 ah[2], al[2], bh[WTN], bl[WTN];
  };
  auto mma = [&](const Frags& f) {
#if DN_MFMA_PRIO
    __builtin_amdgcn_s_setprio(DN_MFMA_PRIO);
#endif
    // (Gray order of the accumulator tiles inside a product group, as conv_sp_kernel's mma: consecutive MFMAs differ in one
    // operand register set; every accumulator receives its three products in the same order -- same bits)
#pragma unroll
    for (int wm = 0; wm < 2; ++wm)
#pragma unroll
      for (int k = 0; k < WTN; ++k) {
        const int wn = (DN_MMA_GRAY && (wm & 1)) ? WTN - 1 - k : k;
        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[wn], f.ah[wm], acc[wm][wn], 0, 0, 0);
      }
#pragma unroll
    for (int wm = 0; wm < 2; ++wm)
#pragma unroll
      for (int k = 0; k < WTN; ++k) {
        const int wn = (DN_MMA_GRAY && (wm & 1)) ? WTN - 1 - k : k;
        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[wn], f.al[wm], acc[wm][wn], 0, 0, 0);
      }
#pragma unroll
    for (int wm = 0; wm < 2; ++wm)
#pragma unroll
      for (int k = 0; k < WTN; ++k) {
        const int wn = (DN_MMA_GRAY && (wm & 1)) ? WTN - 1 - k : k;
        acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[wn], f.ah[wm], acc[wm][wn], 0, 0, 0);
      }
#if DN_MFMA_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  // step ST of a source-1 chunk: taps (dy = ST, dx = 0..2), weights shared by the four waves
  auto compute1 = [&](auto st_c, const unsigned char* As, const unsigned char* Bs) {
    constexpr int DY = decltype(st_c)::value;
    Frags f[2];
    auto load = [&](auto u_c) {
      constexpr int u = decltype(u_c)::value;
      Frags& d = f[u & 1];
#pragma unroll
      for (int wm = 0; wm < 2; ++wm) {
        d.ah[wm] = *reinterpret_cast<const half8*>(As + a_b1[wm][u] + DY * QPITCH * 16);
        d.al[wm] = *reinterpret_cast<const half8*>(As + a_b1[wm][u] + DY * QPITCH * 16 + 2 * QNPIX * 16);
      }
#pragma unroll
      for (int wn = 0; wn < WTN; ++wn) {
        d.bh[wn] = *reinterpret_cast<const half8*>(Bs + b_off[wn] + u * BLK);
        d.bl[wn] = *reinterpret_cast<const half8*>(Bs + b_off[wn] + u * BLK + 2 * BN * 16);
      }
    };
    load(std::integral_constant<int, 0>{});
    load(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    mma(f[0]);
    __builtin_amdgcn_sched_barrier(0);
    load(std::integral_constant<int, 2>{});
    __builtin_amdgcn_sched_barrier(0);
    mma(f[1]);
    __builtin_amdgcn_sched_barrier(0);
    mma(f[0]);
    __builtin_amdgcn_sched_barrier(0);
  };
  // step ST of a source-0 chunk: merged taps t = ST * TGQ + u (a = t >> 1, b = t & 1), this wave's class block
  auto compute0 = [&](auto st_c, const unsigned char* As, const unsigned char* Bs) {
    constexpr int ST = decltype(st_c)::value;
    Frags f[2];
    auto load = [&](auto u_c) {
      constexpr int u = decltype(u_c)::value;
      constexpr int t = ST * TGQ + u, am = t >> 1, b = t & 1;
      Frags& d = f[u & 1];
#pragma unroll
      for (int wm = 0; wm < 2; ++wm) {
        d.ah[wm] = *reinterpret_cast<const half8*>(As + a_b0[wm][b][am]);
        d.al[wm] = *reinterpret_cast<const half8*>(As + a_b0[wm][b][am] + 2 * QNPIX * 16);
      }
#pragma unroll
      for (int wn = 0; wn < WTN; ++wn) {
        d.bh[wn] = *reinterpret_cast<const half8*>(Bs + b_offq[wn] + u * 4 * BLK);
        d.bl[wn] = *reinterpret_cast<const half8*>(Bs + b_offq[wn] + u * 4 * BLK + 2 * BN * 16);
      }
    };
    load(std::integral_constant<int, 0>{});
    if constexpr (TGQ == 4) {
      load(std::integral_constant<int, 1>{});
      __builtin_amdgcn_sched_barrier(0);
      mma(f[0]);
      __builtin_amdgcn_sched_barrier(0);
      load(std::integral_constant<int, 2>{});
      __builtin_amdgcn_sched_barrier(0);
      mma(f[1]);
      __builtin_amdgcn_sched_barrier(0);
      load(std::integral_constant<int, 3>{});
      __builtin_amdgcn_sched_barrier(0);
      mma(f[0]);
      __builtin_amdgcn_sched_barrier(0);
      mma(f[1]);
    } else if constexpr (TGQ == 2) {
      load(std::integral_constant<int, 1>{});
      __builtin_amdgcn_sched_barrier(0);
      mma(f[0]);
      __builtin_amdgcn_sched_barrier(0);
      mma(f[1]);
    } else {
      __builtin_amdgcn_sched_barrier(0);
      mma(f[0]);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- epilogue: affine (+ ReLU) -> split -> permlane gather -> 16-byte SP pieces, as conv_sp_kernel's POST 0.
  // The affine of the current channel block lives in a WAVE-PRIVATE 512-byte LDS block (64 registers at BN = 64
  // would spill): written when the workgroup moves to another block, read by the same wave's epilogue -- in-order
  // LDS operations of one wave, no barrier.  Channels past c_out get scale = shift = 0.
  float* aff_s = reinterpret_cast<float*>(smem + 2 * QA_STAGE + 2 * QB_STAGE) + wave * 128;
  int aff_n0 = -1;
  auto load_affine = [&](int n0) {
    if (n0 == aff_n0) return;
    aff_n0 = n0;
    if (lane < BN) {
      const int co = n0 + lane, ci = min(co, a.c_out - 1);
      aff_s[lane] = co < a.c_out ? a.scale[ci] : 0.f;
      aff_s[64 + lane] = co < a.c_out ? a.shift[ci] : 0.f;
    }
  };
  // Stores: buffer stores against a descriptor of the output image (32-bit lane offset + quarter-plane offset in the
  // vector operand, out-of-map lanes dropped by the bounds check), ReLU as a max against a uniform floor -- as in
  // conv_sp_kernel's store_sp_tile (which also says why the plane offset is not in the scalar operand).
  auto epilogue = [&](const QTile& tc) {
    const int plane = a.h * a.w * 16;
    const int img_bytes = a.cog * 4 * plane;
    const auto rsrc_o = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)tc.img * img_bytes, 0, img_bytes, 0x00020000);
    const float lo_clamp = a.relu ? 0.f : -65504.f;
    if (a.out_f32 != nullptr) {   // fp32 NHWC rows (as conv_sp_kernel's second output): a block of its own behind ONE uniform branch
      const float floor_v = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
      for (int wm = 0; wm < 2; ++wm) {
        const int oy = tc.oy0 + prow[wm], ox = tc.ox0 + pcol;
        const bool inside = oy < a.h && ox < a.w;
        float* orow = a.out_f32 + (((size_t)tc.img * a.h + oy) * a.w + ox) * a.ldo_f32;
#pragma unroll
        for (int wn = 0; wn < WTN; ++wn)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int co = tc.n0 + 32 * wn + 8 * g + 4 * lh;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(aff_s + wn * 32 + 8 * g + 4 * lh);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(aff_s + 64 + wn * 32 + 8 * g + 4 * lh);
            f32x4 v = affine4(quad_of(acc[wm][wn], g), sc, sh);
            nan_seen |= !(fabsf(v[0]) <= 3.4028235e38f) | !(fabsf(v[1]) <= 3.4028235e38f) | !(fabsf(v[2]) <= 3.4028235e38f) |
                        !(fabsf(v[3]) <= 3.4028235e38f);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], floor_v);
            if (inside && co < a.c_out) *reinterpret_cast<f32x4*>(orow + co) = v;
          }
      }
      if (a.out == nullptr) return;      // nothing is split: no magnitude to track
    }
#pragma unroll
    for (int wm = 0; wm < 2; ++wm) {
      const int oy = tc.oy0 + prow[wm], ox = tc.ox0 + pcol;
      const bool inside = oy < a.h && ox < a.w;
      const int voff = inside ? (oy * a.w + ox) * 16 + lh * plane : (int)0x80000000;
#pragma unroll
      for (int wn = 0; wn < WTN; ++wn) {
        u32x2 hi[4], lo[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(aff_s + wn * 32 + 8 * g + 4 * lh);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(aff_s + 64 + wn * 32 + 8 * g + 4 * lh);
          const f32x4 v = affine4(quad_of(acc[wm][wn], g), sc, sh);
          note_nan4_tile(nan_seen, v, g);
          split4(v, hi[g], lo[g], amax, lo_clamp);      // the ReLU rides in the split's clamp (sp_device.h)
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          // chunk (n0 + 32 wn) / 16 + m: lane half 0 ends up with octet 0, lane half 1 with octet 1
          const u32x4 ph = gather_octet(hi[2 * m], hi[2 * m + 1]);
          const u32x4 pl = gather_octet(lo[2 * m], lo[2 * m + 1]);
          const int cg = (tc.n0 + 32 * wn) / 16 + m;
          if (cg < a.cog) {
            __builtin_amdgcn_raw_buffer_store_b128(ph, rsrc_o, voff + cg * 4 * plane, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(pl, rsrc_o, voff + (cg * 4 + 2) * plane, 0, 0);
          }
        }
      }
    }
  };

  // ---- main loop
  int item = blockIdx.x;
  if (item >= a.total_items) return;
  auto partial_of = [&](int j) {     // j = tile * kslices + slice
    return a.ks.partial + ((size_t)j * QNW + wave) * (2 * WTN * 16 * 64) + lane * 4;
  };
  auto fold_acc = [&]() {       // tot += acc (fp32 adds, the order the fix-up pass uses), acc = 0
    if constexpr (KSL != 0) {
#pragma unroll
      for (int wm = 0; wm < 2; ++wm)
#pragma unroll
        for (int wn = 0; wn < WTN; ++wn)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            tot[wm][wn][r] += acc[wm][wn][r];
            acc[wm][wn][r] = 0.f;
          }
    }
  };
  auto zero_tot = [&]() {
    if constexpr (KSL != 0) {
#pragma unroll
      for (int wm = 0; wm < 2; ++wm)
#pragma unroll
        for (int wn = 0; wn < WTN; ++wn)
#pragma unroll
          for (int r = 0; r < 16; ++r) tot[wm][wn][r] = 0.f;
    }
  };
  auto tot_to_acc = [&]() {
    if constexpr (KSL != 0) {
#pragma unroll
      for (int wm = 0; wm < 2; ++wm)
#pragma unroll
        for (int wn = 0; wn < WTN; ++wn) acc[wm][wn] = tot[wm][wn];
    }
  };
  if constexpr (KSL != 0) {
    if (a.ks.fixup) {   // fix-up pass: the split tiles' slices added in slice order from zero, then the epilogue
      for (int p = blockIdx.x; p < a.ks.n_split; p += G) {
        const QTile tc = decode(a.ks.n_whole + p);
        load_affine(tc.n0);
        zero_tot();
        for (int sl = 0; sl < a.ks.count; ++sl) {
          const float* pb = partial_of((p << a.ks.log2) + sl);
#pragma unroll
          for (int wm = 0; wm < 2; ++wm)
#pragma unroll
            for (int wn = 0; wn < WTN; ++wn)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(pb + ((wm * WTN + wn) * 4 + q) * 256);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[wm][wn][4 * q + e] = v[e];
              }
          fold_acc();
        }
        tot_to_acc();
        epilogue(tc);
        note_range(amax, nan_seen);
      }
      return;
    }
  }
  Work cw = decode_work(item);
  QTile cur = cw.tc;
  setup_rsrc(cur);
  setup_voff_b(cur);
  setup_voff_a(cur, first_group(cw) >= a.c0g);
  issue_b(first_group(cw), 0, 0);
  issue_a(first_group(cw), 0);
  int sa = 0, sb = 0;
  bool a_pending = true;   // patch DMAs issued AFTER the weight DMAs the next step waits for

  while (true) {
#pragma unroll
    for (int wm = 0; wm < 2; ++wm)
#pragma unroll
      for (int wn = 0; wn < WTN; ++wn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[wm][wn][r] = 0.f;
    load_affine(cur.n0);
    const bool has_next = item + G < a.total_items;
    QTile nxt = cur;
    int ng0 = 0;                            // first chunk of the next work item
    if (has_next) {
      const Work nw = decode_work(item + G);
      nxt = nw.tc;
      ng0 = first_group(nw);
    }
    if constexpr (KSL != 0) zero_tot();
    const int g_end = KSL ? a.ks.bound(cw.sl1) : nchunks;      // last chunk of the item + 1

    // K slices: one pass of the chunk loop per slice (KSL == 0: one pass over all chunks), the loop body is the same
    for (int sl = cw.sl0; sl < cw.sl1; ++sl) {
    const int g_lo = KSL ? a.ks.bound(sl) : 0, g_hi = KSL ? a.ks.bound(sl + 1) : nchunks;
    for (int g = g_lo; g < g_hi; ++g) {
      const bool last_g = g + 1 == g_end;
      auto step = [&](auto st_c, auto kind_c) {
        constexpr int ST = decltype(st_c)::value;
        constexpr bool SRC1 = decltype(kind_c)::value;
        constexpr int NSG = SRC1 ? NS1 : NS0;
        // This step's weights (and, at ST == 0, this chunk's patch) have landed.  At ST == 1 the NEXT chunk's
        // patch may still be in flight behind them: it was issued after them, so waiting until only this
        // wave's patch instructions remain outstanding is enough.  Raw s_barrier: __syncthreads() would drain vmcnt.
        if (ST == 1 && a_pending) {
          if (wave < QA_INSTR - (QA_IT - 1) * QNW) wait_vm<QA_IT>(); else wait_vm<QA_IT - 1>();
        } else {
          wait_vm0();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ST + 1 < NSG) {
          issue_b(g, ST + 1, sb ^ 1);
        } else if (!last_g) {
          issue_b(g + 1, 0, sb ^ 1);
        } else if (has_next) {
          setup_voff_b(nxt);
          issue_b(ng0, 0, sb ^ 1);
        }
        if (ST == 0) {
          a_pending = true;
          if (!last_g) {
            if (g + 1 == a.c0g) setup_voff_a(cur, true);   // concat: switch to source 1
            issue_a(g + 1, sa ^ 1);
          } else if (has_next) {
            setup_rsrc(nxt);
            setup_voff_a(nxt, ng0 >= a.c0g);
            issue_a(ng0, sa ^ 1);
          } else {
            a_pending = false;
          }
        }
        if constexpr (SRC1 && NS1 == 1) {   // all nine taps in the stage: rows dy = 0..2 of 3 blocks each
          compute1(std::integral_constant<int, 0>{}, smem + sa * QA_STAGE, smem + OFF_B + sb * QB_STAGE);
          compute1(std::integral_constant<int, 1>{}, smem + sa * QA_STAGE, smem + OFF_B + sb * QB_STAGE + 3 * BLK);
          compute1(std::integral_constant<int, 2>{}, smem + sa * QA_STAGE, smem + OFF_B + sb * QB_STAGE + 6 * BLK);
        } else if constexpr (SRC1)
          compute1(st_c, smem + sa * QA_STAGE, smem + OFF_B + sb * QB_STAGE);
        else
          compute0(st_c, smem + sa * QA_STAGE, smem + OFF_B + sb * QB_STAGE);
        sb ^= 1;
      };
      if (g < a.c0g) {
        step(std::integral_constant<int, 0>{}, std::false_type{});
        if constexpr (NS0 >= 2) step(std::integral_constant<int, 1>{}, std::false_type{});
        if constexpr (NS0 == 4) {
          step(std::integral_constant<int, 2>{}, std::false_type{});
          step(std::integral_constant<int, 3>{}, std::false_type{});
        }
      } else {
        step(std::integral_constant<int, 0>{}, std::true_type{});
        if constexpr (NS1 == 3) {
          step(std::integral_constant<int, 1>{}, std::true_type{});
          step(std::integral_constant<int, 2>{}, std::true_type{});
        }
      }
      sa ^= 1;
    }
    if (KSL != 0 && cw.j < 0) fold_acc();      // a slice of a whole tile is complete: add it, restart the chain from zero
    }
    if (KSL != 0 && cw.j >= 0) {
      if constexpr (KSL != 0) {                  // one slice of a split tile: raw accumulators to the partial buffer
        float* pb = partial_of(cw.j);
#pragma unroll
        for (int wm = 0; wm < 2; ++wm)
#pragma unroll
          for (int wn = 0; wn < WTN; ++wn)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *reinterpret_cast<f32x4*>(pb + ((wm * WTN + wn) * 4 + q) * 256) =
                  f32x4{acc[wm][wn][4 * q], acc[wm][wn][4 * q + 1], acc[wm][wn][4 * q + 2], acc[wm][wn][4 * q + 3]};
      }
    } else {
      tot_to_acc();
      epilogue(cur);
      note_range(amax, nan_seen);
    }
    if (!has_next) break;
    item += G;
    if constexpr (KSL != 0) cw = decode_work(item);   // (decoded again rather than carried through the tile: fewer live scalars)
    cur = nxt;
  }
}

// weight_oihw [c_out][c0 + c1][3][3] * wmul -> packed image: chunks of source 0 carry 16 blocks [t = 2 a + b][class
// = py + 2 px], block (t, class) = sum of W[dy][dx] over dy in R(py, a), dx in R(px, b) with R(0, 0) = {0},
// R(0, 1) = {1, 2}, R(1, 0) = {0, 1}, R(1, 1) = {2}; chunks of source 1 the 9 plain taps [dy][dx].
__global__ void spq_pack_weights_kernel(const float* __restrict__ w, unsigned char* __restrict__ wpk, int c_out, int c_in,
                                        int c0g, int cout_pad, int nchunks, float wmul) {
  const long total = ((long)c0g * 16 + (long)(nchunks - c0g) * 9) * 2 * cout_pad;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int n = r % cout_pad; r /= cout_pad;
    const int oct = r % 2; r /= 2;             // r = block index
    int cg, dy0, dy1, dx0, dx1;                // inclusive tap ranges summed
    if (r < (long)c0g * 16) {
      cg = (int)(r / 16);
      const int k = (int)(r % 16), t = k / 4, cls = k % 4;
      const int am = t >> 1, b = t & 1, py = cls & 1, px = cls >> 1;
      dy0 = am == 0 ? 0 : (py == 0 ? 1 : 2); dy1 = am == 0 ? (py == 0 ? 0 : 1) : 2;
      dx0 = b == 0 ? 0 : (px == 0 ? 1 : 2);  dx1 = b == 0 ? (px == 0 ? 0 : 1) : 2;
    } else {
      const long q = r - (long)c0g * 16;
      cg = c0g + (int)(q / 9);
      dy0 = dy1 = (int)(q % 9) / 3;
      dx0 = dx1 = (int)(q % 3);
    }
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ci = cg * 16 + oct * 8 + e;
      float v = 0.f;
      if (n < c_out && ci < c_in) {
        const float* wp = w + ((size_t)n * c_in + ci) * 9;
        for (int dy = dy0; dy <= dy1; ++dy)
          for (int dx = dx0; dx <= dx1; ++dx) v += wp[dy * 3 + dx];
        v *= wmul;
      }
      v = fminf(fmaxf(v, -65504.f), 65504.f);
      hi[e] = (_Float16)v;
      lo[e] = (_Float16)(v - (float)hi[e]);
    }
    unsigned char* d = wpk + (((size_t)r * 4 + oct) * cout_pad + n) * 16;
    *reinterpret_cast<half8*>(d) = hi;
    *reinterpret_cast<half8*>(d + (size_t)2 * cout_pad * 16) = lo;
  }
}

// whole tiles of a K-sliced launch (as conv_sp.hip :: ks_plan): split the tiles of the last, under-filled round
inline long spq_ks_plan(long T, long R, int S, size_t ws_bytes, size_t bytes_per_tile) {
  if (S <= 1 || ws_bytes < bytes_per_tile) return T;
  const long full = T / R, tail = T - full * R;
  if (tail == 0) return T;
  const double cost_split = (double)full + (double)((tail * S + R - 1) / R) / S;
  if (cost_split > (double)(full + 1) - 0.2) return T;
  const long max_split = (long)(ws_bytes / bytes_per_tile);
  return tail > max_split ? T - max_split : full * R;
}

template <int BN, int DEEP = 0, int ABL = 0, int KSL = 0>
int launch_spq(SpqArgs& a, hipStream_t stream, size_t ws_bytes = 0) {
  using T = SpqTile<BN, DEEP>;
  auto kern = conv_spq_kernel<BN, DEEP, ABL, KSL>;
  static dn::PerDeviceFlag attr_flag;
  bool& attr_set = attr_flag.here();
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)T::LDS_BYTES);
    if (e != hipSuccess)
      return dn::fail(DN_ERR_LAUNCH, "spconv (quad-merged): hipFuncSetAttribute(%d B LDS): %s", (int)T::LDS_BYTES,
                      hipGetErrorString(e));
    attr_set = true;
  }
  a.tiles_x = (a.w + QTW - 1) / QTW;
  a.tiles_y = (a.h + QTH - 1) / QTH;
  const long total = (long)a.n_images * a.tiles_y * a.tiles_x * ((a.c_out + BN - 1) / BN);
  DN_REQUIRE(total < (1L << 22), "spconv (quad-merged): too many tiles (%ld)", total);
  DN_REQUIRE((long)a.h * a.w * 16 * 4 * a.cog < (1L << 31), "spconv (quad-merged): one output image exceeds 2 GiB");
  a.total_items = (int)total;
  a.n_cb = (a.c_out + BN - 1) / BN;
  a.rcp_ncb = 1.0f / (float)a.n_cb; a.rcp_tx = 1.0f / (float)a.tiles_x; a.rcp_ty = 1.0f / (float)a.tiles_y;
  const long resident = (DEEP ? 1L : 2L) * kCUs;
  if constexpr (KSL != 0) {
    KSlices& k = a.ks;
    const int S = k.count, ng = a.c0g + a.c1g;
    DN_REQUIRE((S == 2 || S == 4) && ng >= S, "spconv (quad-merged): %d K slices of a %d-chunk layer", S, ng);
    // canonical boundaries: equal shares of the K loop's WORK -- a chunk of the upsampled source runs 4 merged taps, a
    // chunk of the second source 9 -- slice s begins at the first chunk whose preceding work reaches s / S of the total
    const long wtot = 4L * a.c0g + 9L * a.c1g;
    int b[5] = {0, 0, 0, 0, ng};
    for (int sl = 1; sl < S; ++sl) {
      int g = b[sl - 1] + 1;       // at least one chunk per slice
      auto before = [&](int gg) { return gg <= a.c0g ? 4L * gg : 4L * a.c0g + 9L * (gg - a.c0g); };
      while (g < ng - (S - 1 - sl) - 1 && before(g) * S < wtot * sl) ++g;
      b[sl] = g;
    }
    k.log2 = S == 4 ? 2 : 1; k.ngroups = ng;
    k.b1 = b[1]; k.b2 = S == 2 ? ng : b[2]; k.b3 = S == 2 ? ng : b[3];
    const size_t per_tile = (size_t)S * QNW * 2 * (BN / 32) * 16 * 64 * sizeof(float);
    k.n_whole = (int)spq_ks_plan(total, resident, S, k.partial ? ws_bytes : 0, per_tile);
    k.n_split = (int)(total - k.n_whole);
    k.fixup = 0;
    const long work = k.n_whole + (long)k.n_split * S;
    a.total_items = (int)work;
    hipLaunchKernelGGL(kern, dim3((unsigned)(work > resident ? resident : work)), dim3(QNT), T::LDS_BYTES, stream, a);
    if (k.n_split) {
      k.fixup = 1;
      a.total_items = k.n_split;
      // the fix-up pass reads the epilogue affine through the wave-private LDS block: same dynamic LDS
      hipLaunchKernelGGL(kern, dim3((unsigned)(k.n_split > 2 * kCUs ? 2 * kCUs : k.n_split)), dim3(QNT), T::LDS_BYTES, stream, a);
    }
    return dn::check_launch("conv_spq_kernel (K slices)");
  }
  dim3 grid((unsigned)(total > resident ? resident : total));
  hipLaunchKernelGGL(kern, grid, dim3(QNT), T::LDS_BYTES, stream, a);
  return dn::check_launch("conv_spq_kernel");
}

}  // namespace

namespace dn {

void range_collect_conv_spq(unsigned* dst, bool reset, hipStream_t s) { sp_range_collect_here(dst, reset, s); }

size_t spq_packed_blocks(int c0g, int nchunks) { return (size_t)c0g * 16 + (size_t)(nchunks - c0g) * 9; }

int spq_pack_weights(const float* weight_oihw, void* packed, int c_out, int c_in, int c0g, int cout_pad, int nchunks,
                     float wmul, hipStream_t stream) {
  hipLaunchKernelGGL(spq_pack_weights_kernel, dim3(2048), dim3(256), 0, stream, weight_oihw, (unsigned char*)packed, c_out,
                     c_in, c0g, cout_pad, nchunks, wmul);
  return check_launch("spq_pack_weights_kernel");
}

// bn: 32 or 64 output channels per workgroup; 33 = 32 in the one-step-per-chunk (DEEP) form; 0 = choose
int spq_conv(const dn_conv_desc* d, const void* src0, const void* src1, const void* packed, size_t packed_bytes,
             const float* scale, const float* shift, void* out, int cout_pad, int bn, hipStream_t stream,
             int kslices, float* workspace, size_t workspace_bytes, float* out_nhwc, int ld_nhwc) {
  SpqArgs a;
  a.src0 = (const unsigned char*)src0; a.src1 = (const unsigned char*)src1; a.wpk = (const unsigned char*)packed;
  a.scale = scale; a.shift = shift; a.out = (unsigned char*)out;
  a.out_f32 = out_nhwc; a.ldo_f32 = ld_nhwc;
  a.n_images = d->n_images; a.h = d->h_in; a.w = d->w_in;
  a.c0g = (d->c0 + 15) / 16; a.c1g = (d->c1 + 15) / 16;
  a.c_out = d->c_out; a.cog = (d->c_out + 15) / 16; a.relu = d->relu;
  a.cout_pad = cout_pad; a.wpk_bytes = (int)packed_bytes;
  a.tiles_x = a.tiles_y = a.total_items = 0;
  a.ks = KSlices{workspace, kslices, 0, 0, 0, 0, 0, 0, 0, 0};
  if (kslices > 1) {
    // K-sliced layer: the 32-channel tile whatever the launch size (the result must not depend on it); the
    // one-step-per-chunk form when even the slices leave CUs without a workgroup
    const long tiles = (long)d->n_images * ((d->h_in + QTH - 1) / QTH) * ((d->w_in + QTW - 1) / QTW);
    static const int deep_env = [] { const char* e = getenv("DN_SP_DEEP"); return e ? atoi(e) : 1; }();
    const bool deep = deep_env && tiles * ((d->c_out + 31) / 32) * kslices <= (long)kCUs;
    if (bn == 33 || (bn == 0 && deep)) return launch_spq<32, 1, 0, 1>(a, stream, workspace_bytes);
    return launch_spq<32, 0, 0, 1>(a, stream, workspace_bytes);
  }
  if (bn == 0) {
    const long tiles = (long)d->n_images * ((d->h_in + QTH - 1) / QTH) * ((d->w_in + QTW - 1) / QTW);
    // BN = 64 halves the patch traffic per MAC but needs >= two full rounds of 64-wide items (measured: conv7_1
    // 1280 items: 129 vs 153 us; conv6_1 640 items: 146 vs 142 us; conv5_1 320 items: 174 vs 171 us)
    bn = (d->c_out > 32 && tiles * ((d->c_out + 63) / 64) >= 4L * kCUs) ? 64 : 32;
    // fewer items than CUs: latency regime (SpqTile DEEP)
    static const int deep_env = [] { const char* e = getenv("DN_SP_DEEP"); return e ? atoi(e) : 1; }();
    if (deep_env && bn == 32 && tiles * ((d->c_out + 31) / 32) <= (long)kCUs) bn = 33;
  }
  if (bn == 64) return launch_spq<64>(a, stream);
  if (bn == 33) return launch_spq<32, 1>(a, stream);
  if (bn == 101) return launch_spq<32, 0, 1>(a, stream);   // timing-only ablations of the BN = 32 form
  if (bn == 102) return launch_spq<32, 0, 2>(a, stream);
  if (bn == 103) return launch_spq<32, 0, 3>(a, stream);
  return launch_spq<32>(a, stream);
}

}  // namespace dn
