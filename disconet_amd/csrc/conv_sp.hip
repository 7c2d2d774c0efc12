// Split-planar ("SP") implicit-GEMM convolution for gfx950: split-f16 x3 MFMA with
// activations kept PRE-SPLIT in HBM and staged by LDS-DMA.
//
// Why a second conv engine beside conv_mfma.hip: in split-f16 mode that kernel spends more
// time staging than multiplying (fp32 -> hi/lo conversion on the VALU, ds_write of both
// operands, two barriers per 16-channel chunk; MFMA pipe 14-55 % busy,
// profiles/r01_pmc_conv_f16x3.txt).  Here
//   * activations live in HBM as hi/lo f16 planes (sp_layout.h): the producer's epilogue
//     splits once, every consumer reads MFMA-ready fragments -- no VALU work in the loop;
//   * both operands go global -> LDS with `buffer_load_dwordx4 ... lds` (no VGPR round
//     trip, no ds_write); out-of-image halo pieces carry an out-of-range offset and the
//     DMA writes the zero padding;
//   * stages are double-buffered with ONE barrier per step (a step = TG taps of one
//     16-channel chunk, or CA chunks of a 1x1 conv); the activation patch of chunk c+1 is
//     issued a whole chunk ahead, the weights of step s+1 one step ahead;
//   * the LDS images are plain planes [quarter][pixel] x 16 B, so every ds_read_b128 group
//     reads 16 consecutive pieces: conflict-free without padding, for stride 1 and (with
//     de-interleaved columns) stride 2;
//   * the SP epilogue needs no LDS: two v_permlane32_swap per register pair give every lane
//     a complete 16-byte piece, stored as 512-byte runs.
// Arithmetic is that of conv_mfma.hip math mode 1: x = hi + lo, products hi*hi + hi*lo +
// lo*hi on v_mfma_f32_32x32x16_f16, fp32 accumulate, fused affine (+ReLU).
//
// Replaces the conv2d + batch_norm + relu (+ interpolate x2 + cat) chains of
// upstream:coperception/models/det/backbone/Backbone.py :: encode / decode and the heads of
// upstream:coperception/models/det/base/* (SURVEY.md §8 a3, a8, a9).
#ifndef DN_EPI_SOFF
#define DN_EPI_SOFF 0
#endif
#ifndef DN_MFMA_PRIO
#define DN_MFMA_PRIO 0
#endif
#ifndef DN_MMA_GRAY
#define DN_MMA_GRAY 1      // Gray order of the accumulator tiles inside a product group (compute :: mma; 0 = rounds 2-5's row-major order: same bits, conv launches +0.3 %, profiles/r06_mma_gray_ab.txt)
#endif
// tools/ab: 1 = the weight-stationary kernels time their phases with s_memtime (wave 0 of every workgroup, summed into
// g_phase_cycles: [0] wait for the patch + barrier, [1] issue of the next patch, [2] MFMA loop, [3] epilogue, [4] tile decode + rest,
// [5] tiles, [6] total) -- read with dn_sp_phase_cycles().  Never in the shipped build.
#ifndef DN_PHASE_TIMING
#define DN_PHASE_TIMING 0
#endif
#ifndef DN_S2_ABL
#define DN_S2_ABL 0      // tools/ab: 1 = instantiate the timing-only ablations of the stride-2 8 x 8 tile (dn_spconv_force_config(400..407))
#endif
#ifndef DN_UNIFORM_TILE
#define DN_UNIFORM_TILE 0   // tools/ab: 1 = every kernel's tile coordinates through v_readfirstlane (scalar registers)
#endif
#ifndef DN_HEADS_W2_REGS
#define DN_HEADS_W2_REGS 1   // 1: the heads' 1x1 weight fragments held in registers across tiles (shipped); 0: read per tile -- 54 VGPRs fewer, measured 0.5 % slower (round 4, same lease)
#endif
#include "dn_internal.h"
#include "sp_layout.h"
#include "sp_device.h"
#include <cstdlib>
#include <type_traits>

#if DN_PHASE_TIMING
__device__ unsigned long long g_phase_cycles[8];
extern "C" int dn_sp_phase_cycles(unsigned long long* host8, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (host8 && hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_phase_cycles), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#endif

namespace {

constexpr int kCUs = 256;

struct SpArgs {
  const unsigned char* src0;   // SP tensors
  const unsigned char* src1;
  const unsigned char* wpk;    // SP packed weights
  const float* scale;
  const float* shift;
  unsigned char* out;          // SP tensor (or fp32 NHWC for the two-output POST form)
  int n_images, h_in, w_in, h_out, w_out;
  int c0g, c1g;                // 16-channel chunks taken from src0 / src1
  int up0, c_out, cog, relu;   // cog = chunks of the output tensor
  int ngroups;                 // A groups (CA chunks each) of the K loop
  int tiles_x, tiles_y, total_items, xcd_order;
  int n_cb;                    // channel blocks; reciprocals of the work-item decode's divisors (sp_device.h :: fdivmod)
  float rcp_ncb, rcp_tx, rcp_ty;
  int cout_pad, wpk_bytes;
  // fused 1x1 stage
  const unsigned char* w2;     // [2 n-tiles][4 k-steps][2 parts][2 h][32 n] x 16 B
  const float* scale2;
  const float* shift2;
  float* out_b;
  int c_out2, relu2, split2, ldo_a, ldo_b, post_f32;
  int b_total;   // stationary form: bytes of the resident weight block (ngroups * NS * B_STEP)
  int stg_row;   // floats per staged fp32 output row (POST, fp32 out)
  // K slices (KSL kernels; sp_device.h :: KSlices)
  KSlices ks;
  size_t ks_ws_bytes;   // host side: bytes behind ks.partial
};

struct TileCoord {
  int img, oy0, ox0, n0;
};

// BSTAT: weight-stationary form for short-K layers whose whole packed weight block fits the LDS
// beside two patch stages (the full-resolution 32-/64-channel layers): the weights are loaded once
// per workgroup, a step never waits for them, and the only barrier left is the one per 16-channel
// chunk that hands over the patch stage.
// UPM: row-merged form of the 3x3 conv over a nearest-upsampled source (the decoder's *_1 layers).  Patch
// rows 2k+1 and 2k+2 of an upsampled source hold the same low-resolution row, so for an output row of
// parity pi the taps ty = 0, +1 (pi = 0) or ty = -1, 0 (pi = 1) read identical data: their weights are summed
// at pack time and the source's chunks run 2 row-taps x 3 instead of 3 x 3 (-33 % MACs on 2/3 of K).  The
// pixel tile is mapped so that all MFMA tiles of a wave are rows of one parity (parity = wave_m & 1); a step's
// weight stage carries both parities' blocks.
// AHI: source 0 is a HI-ONLY SP tensor ([image][chunk][2 octets][H][W] x 16 B: values that are exact in binary16,
// e.g. the 0/1 occupancy grid): half the patch bytes, no lo fragments, two MFMAs per product instead of three.
// AHI = 2: the same arithmetic from an occupancy BIT grid ([image][H][W] uint32, bit c = channel c, <= 32 channels --
// dn_scatter_dense_bits): the patch's words are loaded to registers and expanded to the hi-only stage's 0x3C00 / 0 halves
// with VALU + ds_write, so the MFMA loop, its operands and every result are those of AHI = 1 on the expanded grid;
// the source is 1/8 of the hi-only bytes (1/32 of the float32 grid).  Weight-stationary form only.
// NB: weight stages of the streaming form -- 2 (a step's weights are requested one step ahead) or 3 (round 5: two steps ahead, for the
// tiles whose steps are much shorter than the loaded L2 latency and whose LDS has room: the stride-2 8 x 8 tile, 78 KB, still two
// workgroups per CU).
template <int KS, int STRIDE, int TH, int TW, int BN, int TG, int CA, int WAVES_M, int WAVES_N,
          int WTM, int WTN, int POST, int BSTAT = 0, int UPM = 0, int AHI = 0, int KSL = 0, int NB = 2>
struct SpTile {
  using P = sp::Patch<KS, STRIDE, TH, TW>;
  static constexpr int NW = WAVES_M * WAVES_N;
  static constexpr int NT = NW * 64;
  static constexpr int TAPS = KS * KS;
  static constexpr int NS = TAPS / TG;          // steps per A group
  static constexpr int SUB = CA * TG;           // (chunk, tap) sub-steps per step
  static constexpr int NPIX = P::NPIX;
  static constexpr int AQ = AHI ? 2 : 4;                    // quarter planes of a source-0 chunk
  static constexpr int A_PIECES = CA * AQ * NPIX;
  static_assert(!AHI || (KS == 3 && CA == 1 && UPM == 0), "hi-only source: 3x3 layers");
  static_assert(AHI != 2 || (BSTAT != 0 && STRIDE == 1), "bit-grid source: weight-stationary stride-1 form");
  // streaming form: every wave issues the same number of DMA instructions per stage (counted
  // vmcnt waits), so a stage is padded to whole rounds of NW instructions; stationary form:
  // only patch DMAs are ever in flight (vmcnt(0) waits), a stage is padded to whole instructions
  static constexpr int A_INSTR = BSTAT ? (A_PIECES + 63) / 64 : ((A_PIECES + NT - 1) / NT) * NW;
  static constexpr int A_IT = (A_INSTR + NW - 1) / NW;
  static constexpr int A_STAGE = A_INSTR * 1024;            // bytes
  static constexpr int B_PIECES = SUB * 4 * BN;
  static constexpr int B_PIECES_MAX = (UPM ? 2 : 1) * B_PIECES;          // merged step: both parities
  static constexpr int B_IT_1 = (B_PIECES + NT - 1) / NT;                // plain step
  static constexpr int B_IT = (B_PIECES_MAX + NT - 1) / NT;
  static constexpr int B_STAGE = B_IT * NT * 16;
  static constexpr int B_STEP = B_PIECES * 16;              // stationary form: steps packed tight
  static_assert(!BSTAT || B_STEP % 1024 == 0, "stationary weights: whole DMA instructions per step");
  static constexpr int RPG = 32 / TW;                       // tile rows per 32-pixel group
  static constexpr int W2_BYTES = POST == 1 ? 2 * 4 * 2 * 2 * 32 * 16 : 0;   // 16 KiB
  static constexpr int STG_ROW = 68;                                    // floats, fp32 staging row (max)
  static constexpr int STG_BYTES = POST == 1 ? NW * 32 * STG_ROW * 4 : 0;
  static constexpr int OFF_B = 2 * A_STAGE;
  static constexpr int OFF_W2 = OFF_B + NB * B_STAGE;       // streaming form (stationary: runtime)
  static_assert(NB == 2 || (NB == 3 && KS == 3 && CA == 1 && TG == 3 && POST == 0 && BSTAT == 0 && UPM == 0 && AHI == 0 && KSL == 0),
                "three weight stages: plain streaming 3x3 tiles, three taps per step");
  static constexpr int OFF_STG = OFF_W2 + W2_BYTES;
  static constexpr int LDS_BYTES = BSTAT ? OFF_B + W2_BYTES : OFF_STG + STG_BYTES;   // stationary (1): + weights
  // stationary form: LDS bytes for `nsteps` weight steps and (POST, fp32 out) staging rows of
  // c_out2 + 4 floats
  static constexpr int lds_stationary(int nsteps, int stg_row) {
    return OFF_B + nsteps * B_STEP + W2_BYTES + (POST == 1 ? NW * 32 * stg_row * 4 : 0);
  }
  static constexpr int OCC_LDS = BSTAT ? (POST == 1 ? 1 : 2) : 160 * 1024 / LDS_BYTES;
  // four accumulator tiles per wave + two fragment sets want > 168 VGPRs: at most 2 workgroups
  // K-sliced form: a second accumulator set (the running sum of the slices) -- two workgroups at two tiles per wave
  static constexpr int OCC_MAX = (WTM * WTN >= 4 || (KSL && WTM * WTN >= 2)) ? 2 : 3;
  static_assert(!KSL || (KS == 3 && CA == 1 && POST == 0 && BSTAT == 0 && UPM == 0 && AHI == 0 && WTM * WTN <= 2),
                "K slices: plain streaming 3x3 tiles with at most two accumulator tiles per wave");
  static constexpr int OCC_W = OCC_LDS < 1 ? 1 : (OCC_LDS > OCC_MAX ? OCC_MAX : OCC_LDS);
  // waves per SIMD the launch bounds promise: NW / 4 per workgroup
  static constexpr int WPS = (OCC_W * NW + 3) / 4;
  static_assert(TAPS % TG == 0, "tap groups must divide the taps");
  static_assert(KS == 3 || (TG == 1 && STRIDE == 1), "1x1: one tap, stride 1");
  static_assert(KS == 1 || CA == 1, "3x3: one chunk per A stage");
  static_assert(TW == 32 || TW == 16 || TW == 8, "tile width");
  static_assert(TW != 8 || (WTM == 1 && TH == 8 && WAVES_M == 2), "8-wide tiles: 2 x 1 pixel groups");
  static_assert(TW == 8 || TH == WAVES_M * WTM * RPG, "pixel tile must match the wave layout");
  static_assert(BN == WAVES_N * WTN * 32, "channel tile must match the wave layout");
  static_assert(!(STRIDE == 2 && TW == 32), "stride 2: 16- or 8-wide tiles");
  static_assert(POST != 1 || (WAVES_N == 1 && BN == 64 && TW == 32), "fused 1x1: 64 channels in one wave");
  static_assert(POST != 2 || (WAVES_N == 1 && (BN == 32 || BN == 64)), "fused block-diagonal 1x1: whole 32-channel heads per wave");
  static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit the LDS");
  static_assert(!UPM || (KS == 3 && STRIDE == 1 && TG == 3 && CA == 1 && POST == 0 && BSTAT == 0 && TH == 8 && WTM == 2 &&
                         ((TW == 32 && WAVES_M == 4) || (TW == 16 && WAVES_M == 2))),
                "row-merged up-conv: 8 x 32 or 8 x 16 streaming tiles");
};

// ABL != 0 builds timing-only ablations for tools/sp_conv_check (never launched by the product):
// 1 = no weight DMA after the first step, 2 = no patch DMA after the first group, 3 = neither,
// 4 = no epilogue stores, 5 = 3 + operands from registers (pure MFMA stream).
template <int KS, int STRIDE, int TH, int TW, int BN, int TG, int CA, int WAVES_M, int WAVES_N,
          int WTM, int WTN, int POST, int ABL = 0, int BSTAT = 0, int UPM = 0, int AHI = 0, int KSL = 0, int NB = 2>
__global__ void __launch_bounds__(
    (WAVES_M * WAVES_N * 64),
    (SpTile<KS, STRIDE, TH, TW, BN, TG, CA, WAVES_M, WAVES_N, WTM, WTN, POST, BSTAT, UPM, AHI, KSL, NB>::WPS))
conv_sp_kernel(const SpArgs a) {
  using T = SpTile<KS, STRIDE, TH, TW, BN, TG, CA, WAVES_M, WAVES_N, WTM, WTN, POST, BSTAT, UPM, AHI, KSL, NB>;
  using P = typename T::P;
  constexpr bool kNoB = ABL == 1 || ABL == 3 || ABL >= 5, kNoA = ABL == 2 || ABL == 3 || ABL >= 5;
  constexpr bool kNoStore = ABL == 4 || ABL == 6 || ABL == 7, kNoLds = ABL >= 5;
  constexpr bool kNoBarrier = ABL == 7;   // 6: + no stores; 7: + no barriers (pure MFMA + epilogue arithmetic)
  constexpr int NW = T::NW, NT = T::NT, NPIX = T::NPIX, TAPS = T::TAPS, NS = T::NS, SUB = T::SUB;
  constexpr int A_IT = T::A_IT, B_IT = T::B_IT;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // POST 1: both stages' affines (<= 64 channels each), staged once: scale, shift, scale2, shift2
  __shared__ __attribute__((aligned(16))) float aff1_s[POST != 0 ? 4 : 1][POST != 0 ? 64 : 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;
  const int li = lane & 31;
  const int lh = lane >> 5;

  // ---- work items (channel block, image, tile_y, tile_x), XCD-aware order as in conv_mfma.hip
  const int G = gridDim.x;
  const int spatial_items = a.n_images * a.tiles_y * a.tiles_x;
  const int n_cb = a.n_cb;
  const int sp_full = spatial_items & ~7;
  auto decode = [&](int item) {
    TileCoord tc;
    int cb, spi;
    if (!a.xcd_order) {
      cb = item / spatial_items;
      spi = item % spatial_items;
    } else if (item < sp_full * n_cb) {
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
    tc.ox0 = tx * TW;
    tc.oy0 = ty * TH;
    if constexpr (AHI == 2 || DN_UNIFORM_TILE) {   // plain buffer loads in the K loop: without this the compiler treats the tile as divergent
      tc.n0 = __builtin_amdgcn_readfirstlane(tc.n0);      // and wraps every buffer store of the epilogue in a waterfall loop
      tc.img = __builtin_amdgcn_readfirstlane(tc.img);
      tc.ox0 = __builtin_amdgcn_readfirstlane(tc.ox0);
      tc.oy0 = __builtin_amdgcn_readfirstlane(tc.oy0);
    }
    return tc;
  };

  // K-sliced launches (KSL): work item v < n_whole is a whole tile (all K slices, folded in registers); the others
  // are single slices of the tiles behind them, `kslices` consecutive work items per tile (sp_device.h :: KSlices)
  // sl0 .. sl1: the slices the item computes (all of them: a whole tile, folded in registers; one: a split tile's
  // slice `sl0`, partial index j = v - n_whole = tile * kslices + slice)
  struct Work {
    TileCoord tc;
    int sl0, sl1, j;
  };
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

  // ---- LDS read offsets (bytes) of this lane's MFMA fragments
  int a_off[WTM], b_off[WTN], prow[WTM], pcol;
  pcol = sp::tile_col<TW>(li);
#pragma unroll
  for (int wm = 0; wm < WTM; ++wm) {
    const int gm = wave_m * WTM + wm;
    prow[wm] = sp::tile_row<TW>(gm, li);
    if constexpr (UPM != 0)   // rows of one parity per wave: TW 32: {w, w + 4}; TW 16: {w + 4 wm, w + 4 wm + 2}
      prow[wm] = wave_m + 4 * wm + (TW == 16 && sp::in_g2(li) ? 2 : 0);
    a_off[wm] = (lh * NPIX + sp::out_base_pos<KS, STRIDE, TH, TW>(prow[wm], pcol)) * 16;
  }
#pragma unroll
  for (int wn = 0; wn < WTN; ++wn) b_off[wn] = (lh * BN + (wave_n * WTN + wn) * 32 + li) * 16;

  f32x16 acc[WTM][WTN];
  f32x16 tot[KSL ? WTM : 1][KSL ? WTN : 1];   // K slices: the sum, in slice order, of the slices' accumulation chains
  float amax = 0.f;   // max |value| this lane has split (range flags, sp_device.h)
  bool nan_seen = false;   // a NaN reached an epilogue (ReLU / the clamp of the split would hide it)

  // ---- DMA state
  constexpr unsigned OOB = 0xFFFFFFFFu;
  const int hs0 = a.up0 ? (a.h_in >> 1) : a.h_in, ws0 = a.up0 ? (a.w_in >> 1) : a.w_in;
  const unsigned plane0 = (unsigned)(hs0 * ws0) * 16u, plane1 = (unsigned)(a.h_in * a.w_in) * 16u;
  const size_t img0_bytes = AHI == 2 ? (size_t)(hs0 * ws0) * 4 : (size_t)a.c0g * T::AQ * plane0;
  const size_t img1_bytes = (size_t)a.c1g * 4 * plane1;
  auto rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.src0), 0, 0, 0x00020000);
  auto rsrc1 = rsrc0;
  const auto rsrcw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.wpk), 0,
                                                        a.wpk_bytes, 0x00020000);
  unsigned voff_a[A_IT], voff_b[B_IT];

  auto opaque = [](int v) {
    asm volatile("" : "+v"(v));
    return v;
  };
  // per-lane source offsets of the A pieces this lane moves: piece (it * NW + wave) * 64 + lane.  Which patch
  // pixel (row r, column cc) and quarter plane cq a piece is does not depend on the tile: decoded ONCE per launch
  // into one packed register per piece (r | cc + 1 << 8 | cq << 16 | valid << 31) -- the two constant divisions and
  // the column map per piece were ~half of the per-tile instruction stream of the short-K layers.
  // A tile then only adds its origin, tests the image bounds and forms the offset.
  unsigned piece_map[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int piece = (it * NW + wave) * 64 + lane;
    const int cq = piece / NPIX, pp = piece % NPIX;   // cq = cu * 4 + q
    const int r = pp / P::PITCH, cc = sp::patch_col_of<KS, STRIDE, TH, TW>(pp % P::PITCH);
    const bool valid = piece < T::A_PIECES && cc >= 0;   // (pad pieces decode to in-range but meaningless fields)
    piece_map[it] = (unsigned)r | ((unsigned)(cc + 1) << 8) | ((unsigned)cq << 16) | (valid ? 0x80000000u : 0u);
  }
  auto setup_voff_a = [&](const TileCoord& tc, bool from1) {
    const int iy0 = tc.oy0 * STRIDE - KS / 2, ix0 = tc.ox0 * STRIDE - KS / 2;
    const unsigned plane = from1 ? plane1 : plane0;
    const int ws = from1 ? a.w_in : ws0;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      unsigned pm = piece_map[it];
      asm volatile("" : "+v"(pm));          // unpack per tile: keeps ONE register per piece live across the K loop
      const int r = pm & 0xff, cc = (int)((pm >> 8) & 0xff) - 1, cq = (pm >> 16) & 0xff;
      const int iy = iy0 + r, ix = ix0 + cc;
      const bool ok = (int)pm < 0 && iy >= 0 && iy < a.h_in && ix >= 0 && ix < a.w_in;
      const int sy = (!from1 && a.up0) ? (iy >> 1) : iy, sx = (!from1 && a.up0) ? (ix >> 1) : ix;
      if constexpr (AHI == 2)
        voff_a[it] = ok ? (unsigned)(sy * ws + sx) * 4u : OOB;       // the pixel's occupancy word
      else
      voff_a[it] = ok ? (unsigned)cq * plane + (unsigned)(sy * ws + sx) * 16u : OOB;
    }
  };
  int voffb_n0 = -1;       // the channel block voff_b was formed for (it only depends on the tile through n0)
  auto setup_voff_b = [&](const TileCoord& tc) {
    if (tc.n0 == voffb_n0) return;
    voffb_n0 = tc.n0;
    const int t = opaque(tid);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int piece = (it * NW + (t >> 6)) * 64 + (t & 63);
      const int u = piece / (4 * BN), q = (piece / BN) % 4, nn = piece % BN;
      const int cu = u / TG, tl = u % TG;
      if constexpr (UPM != 0)   // blocks of a step are consecutive in the packed image (CA = 1)
        voff_b[it] = piece < T::B_PIECES_MAX ? (unsigned)(((u * 4 + q) * a.cout_pad + tc.n0 + nn) * 16) : OOB;
      else
      voff_b[it] = piece < T::B_PIECES
                       ? (unsigned)((((cu * TAPS + tl) * 4 + q) * a.cout_pad + tc.n0 + nn) * 16)
                       : OOB;
    }
  };
  auto setup_rsrc = [&](const TileCoord& tc) {
    rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.src0 + tc.img * img0_bytes),
                                              0, (int)img0_bytes, 0x00020000);
    rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(a.c1g ? a.src1 + tc.img * img1_bytes : a.src0), 0,
        a.c1g ? (int)img1_bytes : 0, 0x00020000);
  };
  // bit-grid source: the occupancy words of this lane's pieces, between issue_a (loads) and commit_a (expand -> LDS)
  unsigned abits[AHI == 2 ? A_IT : 1];
  int abits_g = 0;
  // A group g of the current source set -> stage sa
  auto issue_a = [&](int g, int sa, bool steady = true) {
    if (kNoA && steady) return;
    if constexpr (AHI == 2) {
#pragma unroll
      for (int it = 0; it < A_IT; ++it)      // (pieces past the stage carry the out-of-range offset: no branch around a load)
        abits[it] = __builtin_amdgcn_raw_buffer_load_b32(rsrc0, voff_a[it], 0, 0);   // out of the image: 0
      abits_g = g;
      return;
    }
    const int cg = g * CA;
    const bool from1 = cg >= a.c0g;
    const int soff = from1 ? (cg - a.c0g) * 4 * (int)plane1 : cg * T::AQ * (int)plane0;
    unsigned char* base = smem + sa * T::A_STAGE + wave * 1024;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      if ((it + 1) * NW > T::A_INSTR && it * NW + wave >= T::A_INSTR) break;   // ragged last round (stationary)
      if (from1)
        dma16(rsrc1, base + it * NW * 1024, voff_a[it], soff);
      else
        dma16(rsrc0, base + it * NW * 1024, voff_a[it], soff);
    }
  };
  // bit-grid source: piece (octet cq of chunk abits_g, one pixel) = 8 halves, 0x3C00 where the channel's bit is set --
  // the 16 bytes the hi-only DMA would have delivered, written where it would have written them
  auto commit_a = [&](int sa) {
    if constexpr (AHI == 2) {
      unsigned char* base = smem + sa * T::A_STAGE + (wave * 64 + lane) * 16;
#pragma unroll
      for (int it = 0; it < A_IT; ++it) {
        const unsigned oct = (piece_map[it] >> 16) & 1u;
        const unsigned m = (abits[it] >> (16 * abits_g + 8 * oct)) & 0xffu;
        u32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          v[k] = ((m >> (2 * k)) & 1u ? 0x3C00u : 0u) | ((m >> (2 * k + 1)) & 1u ? 0x3C000000u : 0u);
        if ((it + 1) * NW <= T::A_INSTR || it * NW + wave < T::A_INSTR)       // ragged last round: stay inside the stage
          *reinterpret_cast<u32x4*>(base + it * NW * 1024) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // in LDS before this wave reaches the next barrier
    }
  };
  auto issue_b = [&](int g, int st, int sb, bool steady = true) {
    if (kNoB && steady) return;
    if constexpr (UPM != 0) {
      // packed image: the upsampled source's chunks carry 12 blocks [row-tap 2][parity 2][tx 3], the others 9
      const bool merged = g < a.c0g;
      const int blk = merged ? g * 12 + st * 6 : a.c0g * 12 + (g - a.c0g) * 9 + st * 3;
      const int soff = blk * 4 * a.cout_pad * 16;
      unsigned char* base = smem + T::OFF_B + sb * T::B_STAGE + wave * 1024;
#pragma unroll
      for (int it = 0; it < B_IT; ++it)
        if (merged || it < T::B_IT_1) dma16(rsrcw, base + it * NW * 1024, voff_b[it], soff);
      return;
    }
    const int soff = ((g * CA * TAPS + st * TG) * 4 * a.cout_pad) * 16;
    unsigned char* base = smem + T::OFF_B + sb * T::B_STAGE + wave * 1024;
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
      dma16(rsrcw, base + it * NW * 1024, voff_b[it], soff);
  };

  // ---- MFMAs of sub-steps [U0, U0 + NU) of the tap sequence starting at tap T0: A fragments from
  // the patch stage As, weights from Bs (sub-step u at Bs + u * 4 * BN * 16).  The fragments of
  // sub-step u + 1 are read before the MFMAs of sub-step u are issued (two register sets).
  auto compute = [&](auto t0_c, auto nu_c, const unsigned char* As, const unsigned char* Bs) {
    constexpr int T0 = decltype(t0_c)::value, NU = decltype(nu_c)::value;
    half8 ah[2][WTM], al[2][WTM], bh[2][WTN], bl[2][WTN];
    auto load = [&](auto u_c) {
      constexpr int u = decltype(u_c)::value;
      constexpr int cu = (KS == 1) ? u : 0, tap = (KS == 1) ? 0 : T0 + u;
      constexpr int toff = (cu * 4 * NPIX + sp::tap_offset<KS, STRIDE, TH, TW>(tap / KS, tap % KS)) * 16;
      constexpr int s = u & 1;
      if constexpr (kNoLds) {
        const half8 k = {(_Float16)(li * 1e-3f), (_Float16)0.5f, (_Float16)-0.25f, (_Float16)lh, 0, 0, 0, 0};
#pragma unroll
        for (int wm = 0; wm < WTM; ++wm) ah[s][wm] = al[s][wm] = k;
#pragma unroll
        for (int wn = 0; wn < WTN; ++wn) bh[s][wn] = bl[s][wn] = k;
      } else {
#pragma unroll
        for (int wm = 0; wm < WTM; ++wm) {
          ah[s][wm] = *reinterpret_cast<const half8*>(As + a_off[wm] + toff);
          if constexpr (AHI == 0) al[s][wm] = *reinterpret_cast<const half8*>(As + a_off[wm] + toff + 2 * NPIX * 16);
        }
#pragma unroll
        for (int wn = 0; wn < WTN; ++wn) {
          bh[s][wn] = *reinterpret_cast<const half8*>(Bs + b_off[wn] + u * 4 * BN * 16);
          bl[s][wn] = *reinterpret_cast<const half8*>(Bs + b_off[wn] + (u * 4 + 2) * BN * 16);
        }
      }
    };
    auto mma = [&](auto u_c) {
      constexpr int s = decltype(u_c)::value & 1;
#if DN_MFMA_PRIO
      __builtin_amdgcn_s_setprio(DN_MFMA_PRIO);     // tools/ab: the wave that has its fragments issues MFMAs ahead of a neighbour's epilogue VALU
#endif
      // D[i = channel][j = pixel]; small terms first; independent accumulators interleaved
      // DN_MMA_GRAY (tools/ab): walk the accumulator tiles of a product group in Gray order -- consecutive MFMAs then differ in ONE
      // operand register set, not two (every accumulator still receives its three products in the same order: same bits)
#pragma unroll
      for (int wm = 0; wm < WTM; ++wm)
#pragma unroll
        for (int k = 0; k < WTN; ++k) {
          const int wn = (DN_MMA_GRAY && (wm & 1)) ? WTN - 1 - k : k;
          acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[s][wn], ah[s][wm], acc[wm][wn], 0, 0, 0);
        }
      if constexpr (AHI == 0) {
#pragma unroll
      for (int wm = 0; wm < WTM; ++wm)
#pragma unroll
        for (int k = 0; k < WTN; ++k) {
          const int wn = (DN_MMA_GRAY && (wm & 1)) ? WTN - 1 - k : k;
          acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s][wn], al[s][wm], acc[wm][wn], 0, 0, 0);
        }
      }
#pragma unroll
      for (int wm = 0; wm < WTM; ++wm)
#pragma unroll
        for (int k = 0; k < WTN; ++k) {
          const int wn = (DN_MMA_GRAY && (wm & 1)) ? WTN - 1 - k : k;
          acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s][wn], ah[s][wm], acc[wm][wn], 0, 0, 0);
        }
#if DN_MFMA_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    };
    load(std::integral_constant<int, 0>{});
    auto body = [&](auto u_c) {
      constexpr int u = decltype(u_c)::value;
      // Fences pin "reads of sub-step u + 1, then the MFMAs of sub-step u": left alone, the
      // scheduler sinks every read to just before its first use and the wave stalls on the LDS
      // latency once per sub-step.
      if constexpr (u + 1 < NU) load(std::integral_constant<int, u + 1>{});
      __builtin_amdgcn_sched_barrier(0);
      mma(u_c);
      __builtin_amdgcn_sched_barrier(0);
    };
    body(std::integral_constant<int, 0>{});
    if constexpr (NU > 1) body(std::integral_constant<int, 1>{});
    if constexpr (NU > 2) body(std::integral_constant<int, 2>{});
    if constexpr (NU > 3) body(std::integral_constant<int, 3>{});
    if constexpr (NU > 4) body(std::integral_constant<int, 4>{});
    if constexpr (NU > 5) body(std::integral_constant<int, 5>{});
    if constexpr (NU > 6) body(std::integral_constant<int, 6>{});
    if constexpr (NU > 7) body(std::integral_constant<int, 7>{});
    if constexpr (NU > 8) body(std::integral_constant<int, 8>{});
    static_assert(NU <= 9, "at most 9 sub-steps per compute call");
  };

  // ---- SP epilogue of a 32-pixel x 32-channel accumulator tile: channel tile index ct32 of the
  // output tensor (two 16-channel chunks), pixel (oy, ox); sc/sh indexed by the tile's channels
  // POST 0: the affine of this lane's channels (8 g + 4 lh + e of every 32-channel tile of the block) lives in
  // registers for as long as the workgroup stays on one channel block: 64 dependent global loads per tile and
  // their address arithmetic in front of every epilogue were a third of the short-K layers' time
  constexpr bool kRegAffine = POST == 0;
  f32x4 sc_r[kRegAffine ? WTN : 1][4], sh_r[kRegAffine ? WTN : 1][4];
  int aff_n0 = -1;
#if DN_HEADS_W2_REGS
  half8 w2h[POST == 2 ? WTN : 1][2][2], w2l[POST == 2 ? WTN : 1][2][2];   // POST 2: the heads' 1x1 weights
  int w2_cb[POST == 2 ? WTN : 1];
#pragma unroll
  for (int wn = 0; wn < (POST == 2 ? WTN : 1); ++wn) w2_cb[wn] = -1;
#endif
  auto load_affine = [&](int n0) {
    if (!kRegAffine || n0 == aff_n0) return;
    aff_n0 = n0;
#pragma unroll
    for (int wn = 0; wn < WTN; ++wn)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = n0 + (wave_n * WTN + wn) * 32 + 8 * g + 4 * lh + e;
          const int ci = min(co, a.c_out - 1);            // clamped index + select: no divergent branch
          sc_r[wn][g][e] = co < a.c_out ? a.scale[ci] : 0.f;
          sh_r[wn][g][e] = co < a.c_out ? a.shift[ci] : 0.f;
        }
    // Wait for these loads HERE, with the builtin the compiler's wait-count pass understands.  Left to the pass, their
    // first use -- the epilogue, inside the persistent loop -- gets an s_waitcnt vmcnt(0), which in steady state (no loads
    // pending, the registers long valid) waits for the NEXT tile's patch DMA instead: every epilogue started only after
    // the prefetch it was meant to hide had landed (round 4, phase timing: tools/phase_probe.py).
#if DN_EPI_NO_DMA_WAIT
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
#endif
  };
  // One 32-pixel x 32-channel accumulator tile -> SP pieces.  Stores are buffer stores against a descriptor of the
  // output IMAGE (32-bit lane offset computed once per tile + the quarter-plane offset; a lane outside the map
  // carries an out-of-range offset and the bounds check drops its store): no 64-bit address
  // arithmetic and no exec-mask juggling per store.  ReLU is a max against a uniform floor (0 or -inf).
  auto store_sp_tile = [&](const f32x16& c, const float* scale, const float* shift, int relu,
                           int ch0, int c_lim, unsigned char* out, int cog, int img, int oy, int ox,
                           int wn_r = -1) {
    const bool inside = oy < a.h_out && ox < a.w_out;
    const int plane = a.h_out * a.w_out * 16;                       // bytes of one quarter plane (launch checks the image < 2 GiB)
    const int img_bytes = cog * 4 * plane;
    const auto rsrc_o = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)img * img_bytes, 0, img_bytes, 0x00020000);
    const int voff = inside ? (oy * a.w_out + ox) * 16 + lh * plane : (int)0x80000000;
    // ReLU: in the fp32 copy a max against a uniform floor (0 or -inf); in the split it is the lower clamp bound
    // (sp_device.h :: split4), so the affine for the split applies none (`relu_here` false)
    const float floor_v = relu ? 0.f : -__builtin_inff(), lo_clamp = relu ? 0.f : -65504.f;
    auto affine = [&](int g, f32x4& v, bool relu_here) {
      const int co = ch0 + 8 * g + 4 * lh;
      const float fl = relu_here ? floor_v : -__builtin_inff();
      if (wn_r >= 0) {            // register-resident affine (channels past c_out: scale = shift = 0 -> 0)
        v = affine4(quad_of(c, g), sc_r[kRegAffine ? wn_r : 0][g], sh_r[kRegAffine ? wn_r : 0][g]);
        note_nan4_tile(nan_seen, v, g);
        if (relu_here) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], fl);
        }
      } else if (POST == 1 && wn_r == -2) {   // stage-2 affine of the fused 1x1 from LDS (zero past c_out2)
        const f32x4 sc = lds_table4(reinterpret_cast<const f32x4*>(&aff1_s[POST == 1 ? 2 : 0][co & 63]), smem);
        const f32x4 sh = lds_table4(reinterpret_cast<const f32x4*>(&aff1_s[POST == 1 ? 3 : 0][co & 63]), smem);
        v = affine4(quad_of(c, g), sc, sh);
        note_nan4_tile(nan_seen, v, g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (relu_here) v[e] = fmaxf(v[e], fl);
          v[e] = co + e < c_lim ? v[e] : 0.f;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // clamped index + select: no divergent branch per channel
          const int ci = min(co + e, c_lim - 1);
          v[e] = c[4 * g + e] * scale[ci] + shift[ci];
        }
        note_nan4_tile(nan_seen, v, g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (relu_here) v[e] = fmaxf(v[e], fl);
          v[e] = co + e < c_lim ? v[e] : 0.f;
        }
      }
    };
    if constexpr (POST == 0) {   // optional second output: the same values as fp32 NHWC rows (the exchanged level).
      // A block of its own behind ONE uniform branch: written inside the loop below, every launch without a second
      // output paid a masked-off store sequence per register quad.
      if (a.out_b != nullptr) {
        float* orow = a.out_b + (((size_t)img * a.h_out + oy) * a.w_out + ox) * a.ldo_b;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = ch0 + 8 * g + 4 * lh;
          f32x4 v;
          affine(g, v, false);
          // the fp32 copy leaves the engine (fusion kernels, the agent all-gather): the one place where a NaN / Inf test of
          // the conv stack is free -- one launch per step, behind a uniform branch.  Before the ReLU, which would hide a NaN.
          nan_seen |= !(fabsf(v[0]) <= 3.4028235e38f) | !(fabsf(v[1]) <= 3.4028235e38f) | !(fabsf(v[2]) <= 3.4028235e38f) |
                      !(fabsf(v[3]) <= 3.4028235e38f);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], floor_v);
          if (inside && co < c_lim) *reinterpret_cast<f32x4*>(orow + co) = v;
        }
        // dn_spconv2d_nhwc (the training step's split-f16 data gradient): the fp32 rows are the ONLY output -- nothing is
        // split, no magnitude is tracked (a gradient may exceed the f16 range: it never becomes an f16 pair here)
        if (out == nullptr) return;
      }
    }
    u32x2 hi[4], lo[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v;
      affine(g, v, false);
      split4(v, hi[g], lo[g], amax, lo_clamp);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      // chunk ch0 / 16 + m: lane half 0 ends up with octet 0, lane half 1 with octet 1
      const u32x4 ph = gather_octet(hi[2 * m], hi[2 * m + 1]);
      const u32x4 pl = gather_octet(lo[2 * m], lo[2 * m + 1]);
      const int cg = ch0 / 16 + m;
      if (cg < cog && (!kNoStore || ph[0] == 0x12345678u)) {
        // The plane offset rides in the VECTOR offset, the scalar offset operand stays 0.  With it in the scalar
        // operand hipcc leaves out the wait states between a dwordx4 store and a VALU write of its data registers
        // (its hazard recognizer exempts SGPR offsets; gfx950 needs them): the round-3 form wrote ~1e-4 of the lo
        // pieces wrong, lanes 12-15 / 28-31 of both halves.  DESIGN.md 3.6 (C); tools/soff; tests/test_isa_hazard_cpu.py.
#if DN_EPI_SOFF == 4   // both scalar offsets formed BEFORE the pair: no SALU write of a store's soffset register behind it
        int so_h = __builtin_amdgcn_readfirstlane(cg * 4 * plane), so_l = __builtin_amdgcn_readfirstlane((cg * 4 + 2) * plane);
        asm volatile("" : "+s"(so_h), "+s"(so_l));
        __builtin_amdgcn_raw_buffer_store_b128(ph, rsrc_o, voff, so_h, 0);
        __builtin_amdgcn_raw_buffer_store_b128(pl, rsrc_o, voff, so_l, 0);
#elif DN_EPI_SOFF >= 1   // tools/soff: the round-3 form with the plane offset in the SCALAR operand (1), + wait states behind each store (2, 3)
        __builtin_amdgcn_raw_buffer_store_b128(ph, rsrc_o, voff, cg * 4 * plane, 0);
#if DN_EPI_SOFF == 2
        asm volatile("s_nop 1" ::: "memory");
#elif DN_EPI_SOFF == 3
        asm volatile("s_nop 7" ::: "memory");
#endif
        __builtin_amdgcn_raw_buffer_store_b128(pl, rsrc_o, voff, (cg * 4 + 2) * plane, 0);
#if DN_EPI_SOFF == 2
        asm volatile("s_nop 1" ::: "memory");
#elif DN_EPI_SOFF == 3
        asm volatile("s_nop 7" ::: "memory");
#endif
#else
        __builtin_amdgcn_raw_buffer_store_b128(ph, rsrc_o, voff + cg * 4 * plane, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(pl, rsrc_o, voff + (cg * 4 + 2) * plane, 0, 0);
#endif
      }
    }
  };

  auto epilogue = [&](const TileCoord& tc) {
    if constexpr (POST == 2) {
      // ---- fused BLOCK-DIAGONAL 1x1 stage (the two detection heads): this workgroup's 32 stage-1
      // channels are one head's hidden layer, its 1x1 conv reads nothing else.  Channel block 0 writes
      // columns [0, split2) to `out`, block 1 the rest to `out_b` (fp32 NHWC).  Stage-2 weights come
      // as A fragments straight from L2 (4 KB per head); every lane stores its own 16-byte pieces.
#pragma unroll
      for (int wn = 0; wn < WTN; ++wn) {      // BN == 64: this wave holds both heads' hidden layers
      const int cb = (tc.n0 >> 5) + wn;
      const int c2 = cb ? a.c_out2 - a.split2 : a.split2, c2_0 = cb ? a.split2 : 0;
      float* obase = cb ? a.out_b : reinterpret_cast<float*>(a.out);
      const int ldo = cb ? a.ldo_b : a.ldo_a;
      // stage-2 weight fragments of this head (W2 image: [cb][nt][ks][part][h][32] x 16 B, 4 KB per head, L2-resident):
      // held in registers across tiles (245 VGPRs, no spills, two workgroups per CU either way).  The per-tile read
      // (DN_HEADS_W2_REGS=0: 191 VGPRs) measured 10 us slower per launch in the same lease -- the L2 round trip sits in
      // front of every tile's second stage.
#if DN_HEADS_W2_REGS
      if (cb != w2_cb[wn]) {
        w2_cb[wn] = cb;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const unsigned char* wp = a.w2 + (size_t)((((((cb * 2 + nt) * 2 + ks) * 2 + 0) * 2 + lh) * 32 + li)) * 16;
            w2h[wn][nt][ks] = *reinterpret_cast<const half8*>(wp);
            w2l[wn][nt][ks] = *reinterpret_cast<const half8*>(wp + 2 * 32 * 16);
          }
#if DN_EPI_NO_DMA_WAIT
        __builtin_amdgcn_s_waitcnt(0x0F70);   // as in load_affine: no vmcnt(0) at the fragments' first use in later tiles
#endif
      }
      auto w2_hi = [&](int nt, int ks) { return w2h[wn][nt][ks]; };
      auto w2_lo = [&](int nt, int ks) { return w2l[wn][nt][ks]; };
#else
      half8 w2h_t[2][2], w2l_t[2][2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        if (nt * 32 >= c2) break;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const unsigned char* wp = a.w2 + (size_t)((((((cb * 2 + nt) * 2 + ks) * 2 + 0) * 2 + lh) * 32 + li)) * 16;
          w2h_t[nt][ks] = *reinterpret_cast<const half8*>(wp);
          w2l_t[nt][ks] = *reinterpret_cast<const half8*>(wp + 2 * 32 * 16);
        }
      }
      auto w2_hi = [&](int nt, int ks) { return w2h_t[nt][ks]; };
      auto w2_lo = [&](int nt, int ks) { return w2l_t[nt][ks]; };
#endif
#pragma unroll
      for (int wm = 0; wm < WTM; ++wm) {
        u32x2 hi[4], lo[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = tc.n0 + 32 * wn + 8 * g + 4 * lh;
          const f32x4 sc1 = lds_table4(reinterpret_cast<const f32x4*>(&aff1_s[0][POST == 2 ? co & 63 : 0]), smem);
          const f32x4 sh1 = lds_table4(reinterpret_cast<const f32x4*>(&aff1_s[POST == 2 ? 1 : 0][POST == 2 ? co & 63 : 0]), smem);
          const f32x4 v = affine4(quad_of(acc[wm][wn], g), sc1, sh1);   // the ReLU rides in the split's clamp
          // (no NaN test here: this launch sits at its register budget -- the test cost 30 us of spills -- and its
          // outputs are fp32, where a NaN of the hidden layer that survives the ReLU shows; NaNs of the input were
          // flagged by the epilogue that produced it)
          split4(v, hi[g], lo[g], amax, a.relu ? 0.f : -65504.f);
        }
        half8 xh[2], xl[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          xh[m] = __builtin_bit_cast(half8, gather_octet(hi[2 * m], hi[2 * m + 1]));
          xl[m] = __builtin_bit_cast(half8, gather_octet(lo[2 * m], lo[2 * m + 1]));
        }
        const int oy = tc.oy0 + prow[wm], ox = tc.ox0 + pcol;
        const bool inside = oy < a.h_out && ox < a.w_out;
        float* opx = obase + (((size_t)tc.img * a.h_out + oy) * a.w_out + ox) * ldo;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          if (nt * 32 >= c2) break;
          f32x16 acc2;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const half8 wh = w2_hi(nt, ks), wl = w2_lo(nt, ks);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[ks], acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[ks], acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[ks], acc2, 0, 0, 0);
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int ch = nt * 32 + 8 * g + 4 * lh;        // c2 is a multiple of 4: whole pieces
            if (ch < c2) {
              const f32x4 sc = lds_table4(reinterpret_cast<const f32x4*>(&aff1_s[POST == 2 ? 2 : 0][POST == 2 ? (c2_0 + ch) & 63 : 0]), smem);
              const f32x4 sh = lds_table4(reinterpret_cast<const f32x4*>(&aff1_s[POST == 2 ? 3 : 0][POST == 2 ? (c2_0 + ch) & 63 : 0]), smem);
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[e] = acc2[4 * g + e] * sc[e] + sh[e];
                if (a.relu2) v[e] = fmaxf(v[e], 0.f);
              }
              if (inside && (!kNoStore || v[0] == 12345.678f)) *reinterpret_cast<f32x4*>(opx + ch) = v;
            }
          }
        }
      }
      }
    } else if constexpr (POST == 0) {
#pragma unroll
      for (int wm = 0; wm < WTM; ++wm)
#pragma unroll
        for (int wn = 0; wn < WTN; ++wn)
          store_sp_tile(acc[wm][wn], a.scale, a.shift, a.relu, tc.n0 + (wave_n * WTN + wn) * 32, a.c_out,
                        a.out, a.cog, tc.img, tc.oy0 + prow[wm], tc.ox0 + pcol, wn);
    } else {
      // ---- fused 1x1 stage.  This wave owns all 64 stage-1 channels of its pixels: after the
      // affine + ReLU + split, the permlane gather yields exactly the B-operand fragments
      // (lane (j, h): k = 16 ks + 8 h .. + 7) of stage 2 -- the tile never leaves the registers.
      const unsigned char* W2 = smem + (BSTAT ? T::OFF_B + a.b_total : T::OFF_W2);
      f32x16 acc2[WTM][2];
#pragma unroll
      for (int wm = 0; wm < WTM; ++wm) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc2[wm][nt][r] = 0.f;
#pragma unroll
        for (int wn = 0; wn < 2; ++wn) {
          u32x2 hi[4], lo[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int co = wn * 32 + 8 * g + 4 * lh;
            f32x4 v;
            const f32x4 sc = lds_table4(reinterpret_cast<const f32x4*>(&aff1_s[0][POST == 1 ? co : 0]), smem);
            const f32x4 sh = lds_table4(reinterpret_cast<const f32x4*>(&aff1_s[POST == 1 ? 1 : 0][POST == 1 ? co : 0]), smem);
            v = affine4(quad_of(acc[wm][wn], g), sc, sh);
            note_nan4_tile(nan_seen, v, g);
            split4(v, hi[g], lo[g], amax, a.relu ? 0.f : -65504.f);   // the ReLU rides in the split's clamp
          }
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const int ks = wn * 2 + m;
            const half8 xh = __builtin_bit_cast(half8, gather_octet(hi[2 * m], hi[2 * m + 1]));
            const half8 xl = __builtin_bit_cast(half8, gather_octet(lo[2 * m], lo[2 * m + 1]));
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              // W2 image: [nt][ks][part][h][32] x 16 B
              const half8 wh = *reinterpret_cast<const half8*>(W2 + ((((nt * 4 + ks) * 2 + 0) * 2 + lh) * 32 + li) * 16);
              const half8 wl = *reinterpret_cast<const half8*>(W2 + ((((nt * 4 + ks) * 2 + 1) * 2 + lh) * 32 + li) * 16);
              acc2[wm][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc2[wm][nt], 0, 0, 0);
              acc2[wm][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc2[wm][nt], 0, 0, 0);
              acc2[wm][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc2[wm][nt], 0, 0, 0);
            }
          }
        }
      }
      if (!a.post_f32) {
#pragma unroll
        for (int wm = 0; wm < WTM; ++wm)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            store_sp_tile(acc2[wm][nt], a.scale2, a.shift2, a.relu2, nt * 32, a.c_out2, a.out, a.cog,
                          tc.img, tc.oy0 + prow[wm], tc.ox0 + pcol, -2);
      } else {
        // fp32 NHWC, two outputs: columns [0, split2) -> out (ldo_a), the rest -> out_b (ldo_b).
        // Wave-local staging [32 px][STG_ROW] so that every store instruction is one contiguous run.
        const int stg_row = BSTAT ? a.stg_row : T::STG_ROW;
        float* stg = reinterpret_cast<float*>(smem + (BSTAT ? T::OFF_B + a.b_total + T::W2_BYTES : T::OFF_STG)) +
                     wave * 32 * stg_row;
        const int nc4 = a.c_out2 >> 2, sp4 = a.split2 >> 2;
#pragma unroll
        for (int wm = 0; wm < WTM; ++wm) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int co = nt * 32 + 8 * g + 4 * lh;
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int ci = min(co + e, a.c_out2 - 1);
                v[e] = acc2[wm][nt][4 * g + e] * a.scale2[ci] + a.shift2[ci];
                if (a.relu2) v[e] = fmaxf(v[e], 0.f);
                v[e] = co + e < a.c_out2 ? v[e] : 0.f;
              }
              if (co < stg_row - 4) *reinterpret_cast<f32x4*>(&stg[li * stg_row + co]) = v;
            }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          // TW == 32: the group's 32 pixels are one tile row, consecutive in x
          const int oy = tc.oy0 + wave_m * WTM + wm;
          const size_t px0 = ((size_t)tc.img * a.h_out + oy) * a.w_out + tc.ox0;
          if (oy < a.h_out) {
            for (int idx = lane; idx < 32 * sp4; idx += 64) {
              const int m = idx / sp4, c4 = idx % sp4;
              if (tc.ox0 + m < a.w_out)
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + (px0 + m) * a.ldo_a + 4 * c4) =
                    *reinterpret_cast<const f32x4*>(&stg[m * stg_row + 4 * c4]);
            }
            const int nb4 = nc4 - sp4;
            for (int idx = lane; idx < 32 * nb4; idx += 64) {
              const int m = idx / nb4, c4 = idx % nb4;
              if (tc.ox0 + m < a.w_out)
                *reinterpret_cast<f32x4*>(a.out_b + (px0 + m) * a.ldo_b + 4 * c4) =
                    *reinterpret_cast<const f32x4*>(&stg[m * stg_row + 4 * (sp4 + c4)]);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
    }
  };

  // ---- main loop
  int item = blockIdx.x;
  if (item >= a.total_items) return;
  if constexpr (POST == 2) {   // both stages' affines (64 hidden channels; <= 64 outputs over the two heads)
    if (tid < 64) {
      aff1_s[0][tid] = a.scale[tid];
      aff1_s[POST == 2 ? 1 : 0][tid] = a.shift[tid];
      aff1_s[POST == 2 ? 2 : 0][tid] = tid < a.c_out2 ? a.scale2[tid] : 0.f;
      aff1_s[POST == 2 ? 3 : 0][tid] = tid < a.c_out2 ? a.shift2[tid] : 0.f;
    }
    __syncthreads();
  }
  if constexpr (POST == 1) {
    // stage-2 weights: one linear 16 KiB copy, resident for the whole launch
    for (int i = tid; i < T::W2_BYTES / 16; i += NT)
      *reinterpret_cast<u32x4*>(smem + (BSTAT ? T::OFF_B + a.b_total : T::OFF_W2) + i * 16) =
          *reinterpret_cast<const u32x4*>(a.w2 + i * 16);
    if (tid < 64) {
      aff1_s[0][tid] = a.scale[tid];
      aff1_s[POST == 1 ? 1 : 0][tid] = a.shift[tid];
      aff1_s[POST == 1 ? 2 : 0][tid] = tid < a.c_out2 ? a.scale2[tid] : 0.f;
      aff1_s[POST == 1 ? 3 : 0][tid] = tid < a.c_out2 ? a.shift2[tid] : 0.f;
    }
    __syncthreads();
  }
  TileCoord cur = decode(KSL != 0 ? 0 : item);   // (K-sliced launches decode their first WORK item below)
  setup_rsrc(cur);
  auto zero_acc = [&]() {
#pragma unroll
    for (int wm = 0; wm < WTM; ++wm)
#pragma unroll
      for (int wn = 0; wn < WTN; ++wn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[wm][wn][r] = 0.f;
  };
  int sa = 0;

  if constexpr (BSTAT != 0) {
    // ---- weight-stationary form.  The whole weight block of this layer (one channel block:
    // the launcher guarantees c_out <= BN) goes to LDS once; afterwards only patches move.
    auto load_weights = [&](int n0) {
      const int n_instr = a.b_total >> 10;
      for (int i = wave; i < n_instr; i += NW) {
        const int piece = i * 64 + lane;
        const int step = piece / T::B_PIECES, rem = piece % T::B_PIECES;
        const int u = rem / (4 * BN), q = (rem / BN) % 4, nn = rem % BN;
        const int g = step / NS, st = step % NS, cu = u / TG, tl = u % TG;
        const unsigned idx = (((g * CA + cu) * TAPS + st * TG + tl) * 4 + q) * a.cout_pad + n0 + nn;
        dma16(rsrcw, smem + T::OFF_B + i * 1024, idx * 16u, 0);
      }
    };
    int b_n0 = cur.n0;
#if DN_PHASE_TIMING
    unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = __builtin_readcyclecounter();
    const unsigned long long t_begin = t_prev;
    auto mark = [&](int k) { const unsigned long long t = __builtin_readcyclecounter(); ph[k] += t - t_prev; t_prev = t; };
#else
    auto mark = [&](int) {};
#endif
    load_weights(b_n0);
    setup_voff_a(cur, false);
    issue_a(0, 0, false);
    commit_a(0);
    while (true) {
      zero_acc();
      load_affine(cur.n0);
      const bool has_next = item + G < a.total_items;
      TileCoord nxt = cur;
      if (has_next) nxt = decode(item + G);
      mark(4);
      for (int g = 0; g < a.ngroups; ++g) {
        const bool last_g = g + 1 == a.ngroups;
        wait_vm0();                      // this group's patch (first time: and the weights) has landed ...
        __builtin_amdgcn_s_barrier();    // ... for every wave, and every wave is done with the other stage
        asm volatile("" ::: "memory");
        mark(0);
        if (!last_g) {
          if ((g + 1) * CA == a.c0g && a.c1g) setup_voff_a(cur, true);
          issue_a(g + 1, sa ^ 1);
        } else if (has_next) {
          setup_rsrc(nxt);
          setup_voff_a(nxt, false);
          issue_a(0, sa ^ 1);
        }
        mark(1);
        // all taps of the chunk back to back: nothing in LDS changes under them
        compute(std::integral_constant<int, 0>{}, std::integral_constant<int, NS * SUB>{},
                smem + sa * T::A_STAGE, smem + T::OFF_B + g * (NS * T::B_STEP));
        if (!last_g || has_next) commit_a(sa ^ 1);   // bit-grid source: the words loaded under the MFMAs -> the other stage
        sa ^= 1;
#if DN_PHASE_TIMING
        asm volatile("s_nop 0" ::: "memory");
        {   // the MFMAs are asynchronous: read one accumulator register before the clock (forces the last MFMA to retire)
          float probe = acc[0][0][0];
          asm volatile("v_mov_b32 %0, %0" : "+v"(probe));
        }
#endif
        mark(2);
      }
      epilogue(cur);
      mark(3);
      note_range(amax, nan_seen);
#if DN_PHASE_TIMING
      ph[5] += 1;
#endif
      if (!has_next) break;
      item += G;
      cur = nxt;
      if (cur.n0 != b_n0) {              // another channel block (rare: the launch picks a grid whose
        __builtin_amdgcn_s_barrier();    // stride keeps a workgroup on one block): swap the weights once
        asm volatile("" ::: "memory");   // every wave is done multiplying with the old ones
        b_n0 = cur.n0;
        load_weights(b_n0);
      }
    }
#if DN_PHASE_TIMING
    ph[6] = __builtin_readcyclecounter() - t_begin;
    if (tid == 0)
      for (int k = 0; k < 7; ++k) atomicAdd(&g_phase_cycles[k], ph[k]);
#endif
    return;
  }

  // K slices: partial sums of a split tile <-> global memory, one f32x4 per lane and register quad (1 KB runs)
  auto partial_of = [&](int j) {     // j = tile * kslices + slice
    return a.ks.partial + ((size_t)j * NW + wave) * (WTM * WTN * 16 * 64) + lane * 4;
  };
  auto fold_acc = [&]() {       // tot += acc (fp32 adds, the order the fix-up pass uses), acc = 0
    if constexpr (KSL != 0) {
#pragma unroll
      for (int wm = 0; wm < WTM; ++wm)
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
      for (int wm = 0; wm < WTM; ++wm)
#pragma unroll
        for (int wn = 0; wn < WTN; ++wn)
#pragma unroll
          for (int r = 0; r < 16; ++r) tot[wm][wn][r] = 0.f;
    }
  };
  auto tot_to_acc = [&]() {
    if constexpr (KSL != 0) {
#pragma unroll
      for (int wm = 0; wm < WTM; ++wm)
#pragma unroll
        for (int wn = 0; wn < WTN; ++wn) acc[wm][wn] = tot[wm][wn];
    }
  };
  if constexpr (KSL != 0) {
    if (a.ks.fixup) {
      // ---- fix-up pass: the split tiles' slices, added in slice order from zero -- the arithmetic of fold_acc --
      // then the ordinary epilogue.  No operand DMA, no LDS.
      for (int p = blockIdx.x; p < a.ks.n_split; p += G) {
        const TileCoord tc = decode(a.ks.n_whole + p);
        load_affine(tc.n0);
        zero_tot();
        for (int sl = 0; sl < a.ks.count; ++sl) {
          const float* pb = partial_of((p << a.ks.log2) + sl);
#pragma unroll
          for (int wm = 0; wm < WTM; ++wm)
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
  cur = cw.tc;
  setup_rsrc(cur);
  setup_voff_b(cur);
  setup_voff_a(cur, first_group(cw) * CA >= a.c0g);
  int sb = 0;
  bool a_pending = true;   // A DMAs issued after the B DMAs the next step waits for
  // NB == 3: DMA instructions THIS wave issued behind the weights of the step it will wait for next -- the patch issued two steps
  // ago (hist_a2), the weights issued one step ago (hist_b1), the patch issued one step ago (hist_a1): the counted wait of a step
  int hist_a2 = 0, hist_b1 = 0, hist_a1 = 0;
  if constexpr (NB == 3) {
    issue_a(first_group(cw), 0, false);              // the patch first: the first step waits for it together with its weights
    issue_b(first_group(cw), 0, 0, false);
    issue_b(first_group(cw), 1, 1, false);
    hist_b1 = B_IT;
  } else {
    issue_b(first_group(cw), 0, 0, false);
    issue_a(first_group(cw), 0, false);
  }

  while (true) {
    zero_acc();
    load_affine(cur.n0);
    const bool has_next = item + G < a.total_items;
    TileCoord nxt = cur;
    int ng0 = 0;                            // first K group of the next work item
    if (has_next) {
      const Work nw = decode_work(item + G);
      nxt = nw.tc;
      ng0 = first_group(nw);
    }
    if constexpr (KSL != 0) zero_tot();
    const int g_end = KSL ? a.ks.bound(cw.sl1) : a.ngroups;      // last group of the item + 1

    // K slices: one pass of the group loop per slice (KSL == 0: one pass over all groups), the loop body is the same
    for (int sl = cw.sl0; sl < cw.sl1; ++sl) {
    const int g_lo = KSL ? a.ks.bound(sl) : 0, g_hi = KSL ? a.ks.bound(sl + 1) : a.ngroups;
    for (int g = g_lo; g < g_hi; ++g) {
      const bool last_g = g + 1 == g_end;
      auto step = [&](auto st_c, auto merged_c) {
        constexpr int ST = decltype(st_c)::value;
        constexpr bool MERGED = decltype(merged_c)::value;      // UPM: a row-merged step of the upsampled source
        constexpr int NSG = MERGED ? 2 : NS;                    // steps of this group
        // this step's operands have landed: B (and, at ST == 0, A) were issued one step (one
        // group) ago.  At ST == 1 the A patch of the NEXT group may still be in flight behind B.
        // Raw s_barrier: __syncthreads() would add a fence that drains vmcnt to 0 (the LDS-DMA
        // counts as a pending LDS write) and with it the patch still in flight at ST == 1.
        if constexpr (NB == 3) {
          // this step's weights were requested TWO steps ago; behind them this wave issued hist_a2 + hist_b1 + hist_a1
          // instructions (wave-uniform; every wave issues A_IT / B_IT per stage in the streaming form)
          const int young = hist_a2 + hist_b1 + hist_a1;
          if (young == 0) wait_vm0();
          else if (young == B_IT) wait_vm<B_IT>();
          else if (young == A_IT) wait_vm<A_IT>();
          else wait_vm<A_IT + B_IT>();
        } else {
          if (ST == 1 && a_pending) wait_vm<A_IT>(); else wait_vm0();
        }
        if (!kNoBarrier) __builtin_amdgcn_s_barrier();   // ... for every wave; and every wave is done with the previous step
        asm volatile("" ::: "memory");
        int cnt_b = 0, cnt_a = 0;
        if constexpr (NB == 3) {
          // the weights of the step after next -> the stage the previous step just released
          const int sb2 = sb >= 1 ? sb - 1 : 2;      // (sb + 2) % 3
          if (ST + 2 < NSG) {
            issue_b(g, ST + 2, sb2); cnt_b = B_IT;
          } else if (!last_g) {
            issue_b(g + 1, ST + 2 - NSG, sb2); cnt_b = B_IT;
          } else if (has_next) {
            setup_voff_b(nxt);
            issue_b(ng0, ST + 2 - NSG, sb2); cnt_b = B_IT;
          }
        } else {
        // issue the next step's weights, then (first step of a group) the next group's patch
        if (ST + 1 < NSG) {
          issue_b(g, ST + 1, sb ^ 1);
        } else if (!last_g) {
          issue_b(g + 1, 0, sb ^ 1);
        } else if (has_next) {
          setup_voff_b(nxt);
          issue_b(ng0, 0, sb ^ 1);
        }
        }
        if (ST == 0) {
          a_pending = true;
          if (!last_g) {
            if ((g + 1) * CA == a.c0g && a.c1g) setup_voff_a(cur, true);   // concat: switch source
            issue_a(g + 1, sa ^ 1); cnt_a = A_IT;
          } else if (has_next) {
            setup_rsrc(nxt);
            setup_voff_a(nxt, ng0 * CA >= a.c0g);
            issue_a(ng0, sa ^ 1); cnt_a = A_IT;
          } else {
            a_pending = false;
          }
        }
        hist_a2 = hist_a1; hist_b1 = cnt_b; hist_a1 = cnt_a;
        if constexpr (MERGED) {
          // row-tap ST of an output row of parity pi reads patch row r + ST + (pi & ST); this wave's
          // parity block of the stage's six weight blocks
          const int pi = wave_m & 1;
          compute(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{},
                  smem + sa * T::A_STAGE + (ST + (pi & ST)) * (P::PITCH * 16),
                  smem + T::OFF_B + sb * T::B_STAGE + pi * (3 * 4 * BN * 16));
        } else {
          compute(std::integral_constant<int, ST * TG>{}, std::integral_constant<int, SUB>{},
                  smem + sa * T::A_STAGE, smem + T::OFF_B + sb * T::B_STAGE);
        }
        if constexpr (NB == 3) sb = sb == 2 ? 0 : sb + 1; else sb ^= 1;
      };
      using Plain = std::false_type;
      if (UPM != 0 && g < a.c0g) {
        if constexpr (UPM != 0) {
          step(std::integral_constant<int, 0>{}, std::true_type{});
          step(std::integral_constant<int, 1>{}, std::true_type{});
        }
      } else {
        step(std::integral_constant<int, 0>{}, Plain{});
        if constexpr (NS > 1) step(std::integral_constant<int, 1>{}, Plain{});
        if constexpr (NS > 2) step(std::integral_constant<int, 2>{}, Plain{});
        if constexpr (NS > 3) {
          step(std::integral_constant<int, 3>{}, Plain{}); step(std::integral_constant<int, 4>{}, Plain{});
          step(std::integral_constant<int, 5>{}, Plain{}); step(std::integral_constant<int, 6>{}, Plain{});
          step(std::integral_constant<int, 7>{}, Plain{}); step(std::integral_constant<int, 8>{}, Plain{});
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
        for (int wm = 0; wm < WTM; ++wm)
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

// ---------------------------------------------------------------------------
// layout conversions and weight packing
// ---------------------------------------------------------------------------

// fp32 NHWC [n][h][w][ld] (c channels) -> SP [n][cg][4][h][w] pieces; one thread per piece pair
__global__ void sp_from_nhwc_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst,
                                    int c, int ld, int cg_total, long hw, long total) {
  // idx over (img, cg, oct, pixel): pixel fastest -> coalesced 16-byte stores per plane
  float amax = 0.f;
  bool nan_in = false;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long px = idx % hw;
    long r = idx / hw;
    const int oct = r % 2; r /= 2;
    const int cg = r % cg_total;
    const long img = r / cg_total;
    const float* s = src + (img * hw + px) * ld + cg * 16 + oct * 8;
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = (cg * 16 + oct * 8 + e < c) ? s[e] : 0.f;
      amax = fmaxf(amax, fabsf(x));
      nan_in |= x != x;
      x = fminf(fmaxf(x, -65504.f), 65504.f);
      hi[e] = (_Float16)x;
      lo[e] = (_Float16)(x - (float)hi[e]);
    }
    unsigned char* d = dst + (((img * cg_total + cg) * 4 + oct) * hw + px) * 16;
    *reinterpret_cast<half8*>(d) = hi;
    *reinterpret_cast<half8*>(d + 2 * hw * 16) = lo;
  }
  note_range(amax, nan_in);
}

__global__ void sp_to_nhwc_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst,
                                  int c, int ld, int cg_total, long hw, long total) {
  // idx over (img, pixel, octet): octet fastest -> 32-byte runs per thread, rows contiguous
  const int noct = (c + 7) / 8;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int o = idx % noct;
    const long r = idx / noct;
    const long px = r % hw, img = r / hw;
    const int cg = o / 2, oct = o % 2;
    const unsigned char* s = src + (((img * cg_total + cg) * 4 + oct) * hw + px) * 16;
    const half8 hi = *reinterpret_cast<const half8*>(s);
    const half8 lo = *reinterpret_cast<const half8*>(s + 2 * hw * 16);
    float* d = dst + (img * hw + px) * ld + o * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (o * 8 + e < c) d[e] = (float)hi[e] + (float)lo[e];
  }
}

// weight_oihw [c_out][c_in][k][k] * wmul -> [chunk][tap][q][cout_pad] pieces
// Row-merged image (SpTile UPM) of a 3x3 weight: chunks of the upsampled source (cg < c0g) carry 12 blocks
// [row-tap s][parity pi][tx]: pi = 0: {W[-1], W[0] + W[+1]}, pi = 1: {W[-1] + W[0], W[+1]}; the others 9.
__global__ void sp_pack_weights_up_kernel(const float* __restrict__ w, unsigned char* __restrict__ wpk,
                                          int c_out, int c_in, int c0g, int cout_pad, int nchunks, float wmul) {
  const long total = ((long)c0g * 12 + (long)(nchunks - c0g) * 9) * 2 * cout_pad;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int n = r % cout_pad; r /= cout_pad;
    const int oct = r % 2; r /= 2;             // r = block index
    int cg, ty0, ty1, tx;                      // source rows summed: ty0 (and ty1 when >= 0)
    if (r < (long)c0g * 12) {
      cg = (int)(r / 12);
      const int k = (int)(r % 12), s_ = k / 6, pi = (k / 3) % 2;
      tx = k % 3;
      if (pi == 0) { ty0 = s_ == 0 ? 0 : 1; ty1 = s_ == 0 ? -1 : 2; }
      else         { ty0 = s_ == 0 ? 0 : 2; ty1 = s_ == 0 ? 1 : -1; }
    } else {
      const long q = r - (long)c0g * 12;
      cg = c0g + (int)(q / 9);
      ty0 = (int)(q % 9) / 3; ty1 = -1; tx = (int)(q % 3);
    }
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ci = cg * 16 + oct * 8 + e;
      float v = 0.f;
      if (n < c_out && ci < c_in) {
        const float* wp = w + ((size_t)n * c_in + ci) * 9;
        v = wp[ty0 * 3 + tx];
        if (ty1 >= 0) v += wp[ty1 * 3 + tx];
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

__global__ void sp_pack_weights_kernel(const float* __restrict__ w, unsigned char* __restrict__ wpk,
                                       int c_out, int c_in, int taps, int cout_pad, int nchunks,
                                       float wmul, long total) {
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int n = r % cout_pad; r /= cout_pad;
    const int oct = r % 2; r /= 2;
    const int tap = r % taps;
    const int cg = r / taps;
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ci = cg * 16 + oct * 8 + e;
      float v = (n < c_out && ci < c_in) ? w[((size_t)n * c_in + ci) * taps + tap] * wmul : 0.f;
      v = fminf(fmaxf(v, -65504.f), 65504.f);
      hi[e] = (_Float16)v;
      lo[e] = (_Float16)(v - (float)hi[e]);
    }
    unsigned char* d = wpk + ((((size_t)cg * taps + tap) * 4 + oct) * cout_pad + n) * 16;
    *reinterpret_cast<half8*>(d) = hi;
    *reinterpret_cast<half8*>(d + (size_t)2 * cout_pad * 16) = lo;
  }
}

// ---- every plain-layout pack of a training step in ONE launch (dn_spconv_pack_weights_multi).  A job = one call of
// sp_pack_weights_kernel whose weight tensor W[n][ci][tap] is given by a SOURCE VIEW of a forward weight tensor `w`:
//   mode 0: W[n][ci][t] = w[n][ci_first + ci][t]                              (the layer's own forward; a column cut)
//   mode 1: W[n][ci][t] = w[ci][ci_first + n][taps - 1 - t]                    (dn_conv_dgrad_weights: flipped, transposed cut)
//   mode 2: W[k n_in + j][ci][v] = class (py, px) = (k / 2, k % 2) of dn_conv_dgrad_class_weights over columns ci_first + j
// -- the values the two-launch forms write through a temporary, so the packed images are the same bytes.  A workgroup finds its
// job from the table's first-block column and walks that job's pieces with the single launch's grid-stride loop.
struct PackJob {
  const float* w;
  unsigned char* out;
  int c_out, c_in, taps, cout_pad, nchunks, mode, cin_total, ci_first, n_in, block_first, n_blocks, pad_;
  float wmul, padf_;
  long total;
};
static_assert(sizeof(PackJob) == 80, "dn_spconv_pack_multi_table_bytes");

__device__ inline float pack_job_elem(const PackJob& j, int n, int ci, int tap) {
  if (j.mode == 0) return j.w[((size_t)n * j.cin_total + j.ci_first + ci) * j.taps + tap];
  if (j.mode == 1) return j.w[((size_t)ci * j.cin_total + j.ci_first + n) * j.taps + (j.taps - 1 - tap)];
  const int k = n / j.n_in, jj = n - k * j.n_in, py = k >> 1, px = k & 1, vy = tap / 3, vx = tap - 3 * vy;
  auto src_tap = [](int p, int vv) { return p == 0 ? (vv == 1 ? 1 : -1) : (vv == 1 ? 2 : vv == 2 ? 0 : -1); };
  const int ty = src_tap(py, vy), tx = src_tap(px, vx);
  return (ty < 0 || tx < 0) ? 0.f : j.w[((size_t)ci * j.cin_total + j.ci_first + jj) * 9 + ty * 3 + tx];
}

__global__ void __launch_bounds__(256) sp_pack_weights_multi_kernel(const PackJob* __restrict__ jobs, int n_jobs) {
  __shared__ PackJob job;
  if (threadIdx.x == 0) {
    int lo = 0, hi = n_jobs - 1;                 // the last job whose first block is <= blockIdx.x
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].block_first <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    job = jobs[lo];
  }
  __syncthreads();
  const PackJob& j = job;
  const int vb = blockIdx.x - j.block_first;
  for (long idx = vb * (long)blockDim.x + threadIdx.x; idx < j.total; idx += (long)j.n_blocks * blockDim.x) {
    long r = idx;
    const int n = r % j.cout_pad; r /= j.cout_pad;
    const int oct = r % 2; r /= 2;
    const int tap = r % j.taps;
    const int cg = r / j.taps;
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ci = cg * 16 + oct * 8 + e;
      float v = (n < j.c_out && ci < j.c_in) ? pack_job_elem(j, n, ci, tap) * j.wmul : 0.f;
      v = fminf(fmaxf(v, -65504.f), 65504.f);
      hi[e] = (_Float16)v;
      lo[e] = (_Float16)(v - (float)hi[e]);
    }
    unsigned char* d = j.out + ((((size_t)cg * j.taps + tap) * 4 + oct) * j.cout_pad + n) * 16;
    *reinterpret_cast<half8*>(d) = hi;
    *reinterpret_cast<half8*>(d + (size_t)2 * j.cout_pad * 16) = lo;
  }
}

// w2 [c_out2][c_in2] * wmul -> [nt 2][ks 4][part 2][h 2][n 32] pieces (A-operand fragments)
__global__ void sp_pack_post_kernel(const float* __restrict__ w2, unsigned char* __restrict__ out,
                                    int c_out2, int c_in2, float wmul) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // (nt, ks, h, n)
  if (idx >= 2 * 4 * 2 * 32) return;
  const int n = idx % 32, h = (idx / 32) % 2, ks = (idx / 64) % 4, nt = idx / 256;
  half8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int row = nt * 32 + n, k = ks * 16 + h * 8 + e;
    float v = (row < c_out2 && k < c_in2) ? w2[(size_t)row * c_in2 + k] * wmul : 0.f;
    v = fminf(fmaxf(v, -65504.f), 65504.f);
    hi[e] = (_Float16)v;
    lo[e] = (_Float16)(v - (float)hi[e]);
  }
  *reinterpret_cast<half8*>(out + ((((nt * 4 + ks) * 2 + 0) * 2 + h) * 32 + n) * 16) = hi;
  *reinterpret_cast<half8*>(out + ((((nt * 4 + ks) * 2 + 1) * 2 + h) * 32 + n) * 16) = lo;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
inline int out_dim(int in, int ksize, int stride) {
  const int pad = ksize / 2;
  return (in + 2 * pad - ksize) / stride + 1;
}
inline int cout_pad_of(int c_out) { return (c_out + 63) / 64 * 64; }
inline int chunks_of(int c) { return (c + 15) / 16; }
// weights are padded to whole groups of 4 chunks so that a 1x1 A group never runs past them
// Row-merged form of a 3x3 conv whose first source is nearest-upsampled (SpTile UPM): decided by the layer
// alone, so that pack time and run time agree (DN_SP_UPMERGE=0 switches it off for the process).
// -> 0: plain taps; 1: row-merged (UPM, this file); 2: row- and column-merged per parity class (conv_spq.hip, the
// default).  DN_SP_UPMERGE=0|1|2 / dn_spconv_set_upmode() choose for the process: set before the first pack.
int g_sp_upmode = -1;
inline int up_mode(const dn_conv_desc& d) {
  static const int env = [] { const char* e = getenv("DN_SP_UPMERGE"); return e ? atoi(e) : 2; }();
  const int mode = g_sp_upmode >= 0 ? g_sp_upmode : env;
  const bool ok = d.up0 == 1 && d.ksize == 3 && d.stride == 1 && d.c0 > 0 && d.c0 % 16 == 0 && d.h_in % 2 == 0 &&
                  d.w_in % 2 == 0;
  return ok ? (mode < 0 ? 0 : mode > 2 ? 2 : mode) : 0;
}
inline bool up_merged(const dn_conv_desc& d) { return up_mode(d) == 1; }
inline size_t packed_blocks(const dn_conv_desc& d);
inline int packed_chunks(const dn_conv_desc& d) { return (chunks_of(d.c0) + chunks_of(d.c1) + 3) / 4 * 4; }

int validate(const dn_conv_desc* d) {
  DN_REQUIRE(d != nullptr, "spconv: null descriptor");
  DN_REQUIRE(d->ksize == 1 || d->ksize == 3, "spconv: ksize %d unsupported (1 or 3)", d->ksize);
  DN_REQUIRE(d->stride == 1 || (d->stride == 2 && d->ksize == 3),
             "spconv: stride %d with ksize %d unsupported", d->stride, d->ksize);
  DN_REQUIRE(d->n_images > 0 && d->h_in > 0 && d->w_in > 0, "spconv: empty input");
  DN_REQUIRE(d->c0 > 0 && d->c1 >= 0 && d->c_out > 0, "spconv: bad channel counts");
  DN_REQUIRE(d->up0 == 0 || d->up0 == 1, "spconv: up0 must be 0 or 1");
  DN_REQUIRE(d->math != 3 || (d->ksize == 3 && d->stride == 1 && d->c1 == 0 && d->up0 == 0),
             "spconv: a hi-only source 0 (math = 3) needs a 3x3 stride-1 single-source layer");
  DN_REQUIRE(d->math != 4 || (d->ksize == 3 && d->stride == 1 && d->c1 == 0 && d->up0 == 0 && d->c0 <= 32),
             "spconv: a bit-grid source 0 (math = 4) needs a 3x3 stride-1 single-source layer of <= 32 channels");
  DN_REQUIRE(!d->up0 || (d->h_in % 2 == 0 && d->w_in % 2 == 0), "spconv: x2-upsampled source needs even h_in/w_in");
  DN_REQUIRE(d->c1 == 0 || (d->c0 % 16 == 0 && d->ksize == 3), "spconv: concat needs 3x3 and c0 %% 16 == 0 (c0 = %d)", d->c0);
  const size_t hs0 = d->up0 ? d->h_in / 2 : d->h_in, ws0 = d->up0 ? d->w_in / 2 : d->w_in;
  DN_REQUIRE(hs0 * ws0 * chunks_of(d->c0) * 64 < (1ull << 31) &&
                 (size_t)d->h_in * d->w_in * chunks_of(d->c1) * 64 < (1ull << 31),
             "spconv: one image must stay below 2 GiB");
  return DN_OK;
}

inline size_t packed_blocks(const dn_conv_desc& d) {   // 16-byte-piece blocks of [4 quarters][cout_pad] in the packed image
  const int nch = packed_chunks(d);
  if (up_mode(d) == 2) return dn::spq_packed_blocks(chunks_of(d.c0), nch);
  if (up_merged(d)) return (size_t)chunks_of(d.c0) * 12 + (size_t)(nch - chunks_of(d.c0)) * 9;
  return (size_t)nch * d.ksize * d.ksize;
}

enum SpCfgId { S3_256x64, S3_256x32, S3_128x64, S3_64x64, S3S2_128x64, S3S2_64x64, S1_256x64, S1_64x64,
               S3_256x64_T9, S3_512x64, S3_256x128, S1_256x64_C1, S3_256x32_ST, S3_64x64_T9, S3_128x64_T9,
               S3S2_64x64_T9, S3S2_128x64_T9, SP_CFG_COUNT };
// chunks per A stage of the 1x1 tiles: the chunk count of the input must be a multiple
inline int ca_of(SpCfgId id) { return id == S1_256x64 ? 2 : id == S1_64x64 ? 4 : 1; }
struct SpCfg { SpCfgId id; int th, tw, bn; float bias; };
// biases: measured time per unit of tile area relative to 256x64 (tools/sp_conv_check.hip sweep)
float g_sp_bias[SP_CFG_COUNT] = {1.00f, 1.15f, 1.10f, 1.40f, 1.45f, 1.00f, 1.00f, 1.30f, 1.f, 1.f, 1.f, 1.2f,
                                 1.f, 1.f, 1.f, 1.f, 1.f};
const SpCfg kSpCfgs[SP_CFG_COUNT] = {
    {S3_256x64, 8, 32, 64, 0},   {S3_256x32, 8, 32, 32, 0},   {S3_128x64, 8, 16, 64, 0},
    {S3_64x64, 8, 8, 64, 0},     {S3S2_128x64, 8, 16, 64, 0}, {S3S2_64x64, 8, 8, 64, 0},
    {S1_256x64, 8, 32, 64, 0},   {S1_64x64, 8, 8, 64, 0},
    {S3_256x64_T9, 8, 32, 64, 0}, {S3_512x64, 16, 32, 64, 0}, {S3_256x128, 8, 32, 128, 0},
    {S1_256x64_C1, 8, 32, 64, 0}, {S3_256x32_ST, 8, 32, 32, 0},
    {S3_64x64_T9, 8, 8, 64, 0},  {S3_128x64_T9, 8, 16, 64, 0}, {S3S2_64x64_T9, 8, 8, 64, 0},
    {S3S2_128x64_T9, 8, 16, 64, 0},
};
// Launches that leave most CUs with one workgroup or none (the deep layers of a 4-image agent share) run at the
// latency of one K step, not at MFMA throughput: the all-nine-taps-per-step variant of the same tile has a third
// of the steps (one 36 KB weight stage per 16-channel chunk instead of three 12 KB ones)
inline SpCfgId deep_variant(SpCfgId id) {
  switch (id) {
    case S3_256x64: return S3_256x64_T9;
    case S3_128x64: return S3_128x64_T9;
    case S3_64x64: return S3_64x64_T9;
    case S3S2_64x64: return S3S2_64x64_T9;
    case S3S2_128x64: return S3S2_128x64_T9;
    default: return id;
  }
}
int g_sp_force = -1;   // tools: force one configuration

SpCfg select_cfg(const dn_conv_desc& d, int kslices = 1, bool can_split = false) {
  const int ho = out_dim(d.h_in, d.ksize, d.stride), wo = out_dim(d.w_in, d.ksize, d.stride);
  static const SpCfgId c3[] = {S3_256x64, S3_256x32, S3_128x64, S3_64x64};   // merged layers: the first three
  static const SpCfgId c3s2[] = {S3S2_128x64, S3S2_64x64};
  static const SpCfgId c1[] = {S1_256x64, S1_64x64, S1_256x64_C1};
  // row-merged layers: the 256x64 tile's doubled weight stage leaves one workgroup per CU (201 vs 180 us on conv7_1)
  static const SpCfgId c3up[] = {S3_256x32, S3_128x64};
  const bool upm = up_merged(d);
  // K-sliced layers: tiles with at most two accumulator tiles per wave (the slices' running sum is a second set)
  static const SpCfgId c3ks[] = {S3_256x32, S3_128x64, S3_64x64};
  const bool ksl = kslices > 1 && d.ksize == 3 && !upm;
  const SpCfgId* cand = d.ksize == 1 ? c1 : (d.stride == 2 ? c3s2 : (upm ? c3up : (ksl ? c3ks : c3)));
  const int ncand = d.ksize == 1 ? 3 : (d.stride == 2 ? 2 : (upm ? 2 : (ksl ? 3 : 4)));
  const int nchunks = chunks_of(d.c0) + chunks_of(d.c1);
  if (g_sp_force >= 0) {
    for (int k = 0; k < ncand; ++k)
      if (cand[k] == g_sp_force && nchunks % ca_of(cand[k]) == 0) return kSpCfgs[cand[k]];
    if (d.ksize == 3 && d.stride == 1 && !upm && g_sp_force >= S3_256x64_T9 && g_sp_force <= S3_128x64_T9 &&
        g_sp_force != S1_256x64_C1)
      return kSpCfgs[g_sp_force];
    if (d.ksize == 3 && d.stride == 2 && (g_sp_force == S3S2_64x64_T9 || g_sp_force == S3S2_128x64_T9))
      return kSpCfgs[g_sp_force];
  }
  SpCfg best = kSpCfgs[cand[0]];
  double best_cost = 1e300;
  long best_blocks = 0;
  for (int k = 0; k < ncand; ++k) {
    const SpCfg& c = kSpCfgs[cand[k]];
    if (nchunks % ca_of(c.id) != 0) continue;
    const long tiles = (long)d.n_images * ((ho + c.th - 1) / c.th) * ((wo + c.tw - 1) / c.tw);
    const long blocks = tiles * ((d.c_out + c.bn - 1) / c.bn);
    // a K-sliced launch fills its last round with slices: rounds in units of 1 / kslices
    const double rounds = ksl && can_split ? (double)((blocks * kslices + kCUs - 1) / kCUs) / kslices
                              : (double)((blocks + kCUs - 1) / kCUs);
    const double cost = rounds * c.th * c.tw * c.bn * g_sp_bias[c.id];
    if (cost < best_cost * 0.999) { best_cost = cost; best = c; best_blocks = blocks; }
  }
  static const int deep_env = [] { const char* e = getenv("DN_SP_DEEP"); return e ? atoi(e) : 1; }();
  if (deep_env && g_sp_force < 0 && !upm && best_blocks * (ksl && can_split ? kslices : 1) <= kCUs) best = kSpCfgs[deep_variant(best.id)];
  return best;
}

// Which tiles of a K-sliced launch are split (sp_device.h :: KSlices).  T whole-tile items on R resident workgroups run
// floor(T / R) full rounds and a last one that leaves CUs idle (or, T < R, never fills the chip): the tiles of that
// round are handed out slice by slice when that shortens the launch by a fifth of a round or more, as far as the
// workspace reaches.  -> number of whole tiles.
inline long ks_plan(long T, long R, int S, size_t ws_bytes, size_t bytes_per_tile) {
  if (S <= 1 || ws_bytes < bytes_per_tile) return T;
  const long full = T / R, tail = T - full * R;
  if (tail == 0) return T;
  const double cost_split = (double)full + (double)((tail * S + R - 1) / R) / S;
  if (cost_split > (double)(full + 1) - 0.2) return T;
  const long max_split = (long)(ws_bytes / bytes_per_tile);
  return tail > max_split ? T - max_split : full * R;
}

template <int KS, int STRIDE, int TH, int TW, int BN, int TG, int CA, int WAVES_M, int WAVES_N,
          int WTM, int WTN, int POST = 0, int ABL = 0, int BSTAT = 0, int UPM = 0, int AHI = 0, int KSL = 0, int NB = 2>
int launch(SpArgs& a, const dn_conv_desc& d, hipStream_t stream) {
  using T = SpTile<KS, STRIDE, TH, TW, BN, TG, CA, WAVES_M, WAVES_N, WTM, WTN, POST, BSTAT, UPM, AHI, KSL, NB>;
  auto kern = conv_sp_kernel<KS, STRIDE, TH, TW, BN, TG, CA, WAVES_M, WAVES_N, WTM, WTN, POST, ABL, BSTAT, UPM, AHI, KSL, NB>;
  const int nchunks = a.c0g + a.c1g;
  DN_REQUIRE(nchunks % CA == 0, "spconv: chunk count %d not a multiple of %d", nchunks, CA);
  a.ngroups = nchunks / CA;
  int lds_bytes = T::LDS_BYTES;
  constexpr int kStaticLds = POST != 0 ? 1024 : 0;   // aff1_s of the fused-1x1 kernel (static, on top of the dynamic block)
  if (BSTAT) {
    a.b_total = a.ngroups * T::NS * T::B_STEP;
    a.stg_row = a.c_out2 + 4;
    lds_bytes = T::lds_stationary(a.ngroups * T::NS, a.post_f32 ? a.stg_row : 0);
    DN_REQUIRE(lds_bytes <= 160 * 1024 - kStaticLds && (d.c_out <= BN || POST == 2),
               "spconv: layer does not fit the weight-stationary form");
  }
  // opt in to > 64 KiB of dynamic LDS; the attribute write is idempotent, so two first
  // callers racing here only repeat it
  static dn::PerDeviceFlag attr_flag;
  bool& attr_set = attr_flag.here();
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       BSTAT ? 160 * 1024 - kStaticLds : (int)T::LDS_BYTES);
    if (e != hipSuccess)
      return dn::fail(DN_ERR_LAUNCH, "spconv: hipFuncSetAttribute(%d B LDS): %s", (int)T::LDS_BYTES,
                      hipGetErrorString(e));
    attr_set = true;
  }
  // resident workgroups per CU: what the LDS admits, capped by the register budget the
  // launch bounds were compiled for
  int occupancy = 160 * 1024 / lds_bytes;
  occupancy = occupancy < 1 ? 1 : (occupancy > T::OCC_W ? T::OCC_W : occupancy);
  a.tiles_x = (a.w_out + TW - 1) / TW;
  a.tiles_y = (a.h_out + TH - 1) / TH;
  const long total = (long)a.n_images * a.tiles_y * a.tiles_x * ((d.c_out + BN - 1) / BN);
  DN_REQUIRE(total < (1L << 31), "spconv: too many tiles (%ld)", total);
  DN_REQUIRE(total < (1L << 22), "spconv: too many work items for the reciprocal decode (%ld)", total);
  a.total_items = (int)total;
  a.xcd_order = 1;
  a.n_cb = (d.c_out + BN - 1) / BN;
  a.rcp_ncb = 1.0f / (float)a.n_cb; a.rcp_tx = 1.0f / (float)a.tiles_x; a.rcp_ty = 1.0f / (float)a.tiles_y;
  const long resident = (long)occupancy * kCUs;
  if constexpr (KSL != 0) {
    const int S = a.ks.count;
    DN_REQUIRE(S == 2 || S == 4, "spconv: %d K slices (1, 2 or 4)", S);
    DN_REQUIRE(a.ngroups >= S, "spconv: %d K slices of a %d-chunk layer", S, a.ngroups);
    KSlices& k = a.ks;
    k.count = S; k.log2 = S == 4 ? 2 : 1; k.ngroups = a.ngroups;
    k.b1 = a.ngroups / S; k.b2 = 2 * a.ngroups / S; k.b3 = 3 * a.ngroups / S;   // equal shares of the chunks
    if (S == 2) { k.b2 = k.b3 = a.ngroups; }
    const size_t per_tile = (size_t)S * T::NW * WTM * WTN * 16 * 64 * sizeof(float);
    k.n_whole = (int)ks_plan(total, resident, S, k.partial ? a.ks_ws_bytes : 0, per_tile);
    k.n_split = (int)(total - k.n_whole);
    k.fixup = 0;
    const long work = k.n_whole + (long)k.n_split * S;
    a.total_items = (int)work;
    hipLaunchKernelGGL(kern, dim3((unsigned)(work > resident ? resident : work)), dim3(T::NT), lds_bytes, stream, a);
    if (k.n_split) {     // the split tiles' slices added in slice order + their epilogue: same kernel, no operands
      k.fixup = 1;
      a.total_items = k.n_split;
      hipLaunchKernelGGL(kern, dim3((unsigned)(k.n_split > 4 * kCUs ? 4 * kCUs : k.n_split)), dim3(T::NT), 0, stream, a);
    }
    return dn::check_launch("conv_sp_kernel (K slices)");
  }
  dim3 grid((unsigned)(total > resident ? resident : total));
  hipLaunchKernelGGL(kern, grid, dim3(T::NT), lds_bytes, stream, a);
  return dn::check_launch("conv_sp_kernel");
}

// does the layer fit the weight-stationary form of tile <BN, TG> beside `wgs` workgroups per CU?
inline bool fits_stationary(const dn_conv_desc& d, int bn, int a_stage, int extra, int wgs,
                            bool any_blocks = false) {
  if (d.ksize != 3 || d.stride != 1 || (d.c_out > bn && !any_blocks)) return false;
  const int nchunks = chunks_of(d.c0) + chunks_of(d.c1);
  const int bytes = 2 * a_stage + nchunks * 9 * 4 * bn * 16 + extra;
  return bytes * wgs <= 160 * 1024;
}

int fill_args(const dn_conv_desc* d, const void* src0, const void* src1, const void* packed,
              const float* scale, const float* shift, void* out, SpArgs& a) {
  a.src0 = (const unsigned char*)src0; a.src1 = (const unsigned char*)src1;
  a.wpk = (const unsigned char*)packed; a.scale = scale; a.shift = shift; a.out = (unsigned char*)out;
  a.n_images = d->n_images; a.h_in = d->h_in; a.w_in = d->w_in;
  a.h_out = out_dim(d->h_in, d->ksize, d->stride);
  a.w_out = out_dim(d->w_in, d->ksize, d->stride);
  a.c0g = chunks_of(d->c0); a.c1g = chunks_of(d->c1);
  a.up0 = d->up0; a.c_out = d->c_out; a.cog = chunks_of(d->c_out); a.relu = d->relu;
  a.cout_pad = cout_pad_of(d->c_out);
  a.wpk_bytes = (int)(packed_blocks(*d) * 4 * a.cout_pad * 16);
  // the epilogue's buffer stores address one output image with 32-bit offsets (64 is the widest fused second stage)
  DN_REQUIRE((long)a.h_out * a.w_out * 16 * 4 * (a.cog > 4 ? a.cog : 4) < (1L << 31),
             "spconv: one output image of %d x %d x %d channels exceeds 2 GiB", a.h_out, a.w_out, d->c_out);
  a.w2 = nullptr; a.scale2 = nullptr; a.shift2 = nullptr; a.out_b = nullptr;
  a.c_out2 = 0; a.relu2 = 0; a.split2 = 0; a.ldo_a = 0; a.ldo_b = 0; a.post_f32 = 0;
  a.b_total = 0; a.stg_row = 0;
  a.ks = KSlices{nullptr, 1, 0, 0, 0, 0, 0, 0, 0, 0};
  a.ks_ws_bytes = 0;
  auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  DN_REQUIRE(aligned16(src0) && aligned16(src1) && aligned16(packed) && aligned16(out),
             "spconv: tensors must be 16-byte aligned");
  return DN_OK;
}

}  // namespace

namespace dn { void range_collect_conv_sp(unsigned* dst, bool reset, hipStream_t s) { sp_range_collect_here(dst, reset, s); } }
// device address (current device) of this unit's flag word: kernels of OTHER units that split values (the BatchNorm backward's SP
// copy of dz, train_ops.hip) report into it instead of growing the list of collectors a captured step replays
namespace dn {
unsigned* sp_range_word() {
  void* p = nullptr;
  return hipGetSymbolAddress(&p, HIP_SYMBOL(g_sp_range_flags)) == hipSuccess ? (unsigned*)p : nullptr;
}
}

extern "C" size_t dn_sp_tensor_bytes(int n_images, int h, int w, int channels) {
  return (size_t)n_images * chunks_of(channels) * 4 * h * w * 16;
}

extern "C" int dn_sp_from_nhwc(const float* src, int n_images, int h, int w, int channels, int ld,
                               void* dst, void* stream) {
  DN_REQUIRE(src && dst && n_images > 0 && h > 0 && w > 0 && channels > 0 && ld >= channels,
             "sp_from_nhwc: bad arguments");
  const int cg = chunks_of(channels);
  const long hw = (long)h * w, total = (long)n_images * cg * 2 * hw;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(sp_from_nhwc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src,
                     (unsigned char*)dst, channels, ld, cg, hw, total);
  return dn::check_launch("sp_from_nhwc_kernel");
}

extern "C" int dn_sp_to_nhwc(const void* src, int n_images, int h, int w, int channels, int ld,
                             float* dst, void* stream) {
  DN_REQUIRE(src && dst && n_images > 0 && h > 0 && w > 0 && channels > 0 && ld >= channels,
             "sp_to_nhwc: bad arguments");
  const int cg = chunks_of(channels);
  const long hw = (long)h * w, total = (long)n_images * hw * ((channels + 7) / 8);
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(sp_to_nhwc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)src, dst, channels, ld, cg, hw, total);
  return dn::check_launch("sp_to_nhwc_kernel");
}

extern "C" size_t dn_spconv_packed_weight_bytes(const dn_conv_desc* d) {
  if (validate(d) != DN_OK) return 0;
  return packed_blocks(*d) * 4 * cout_pad_of(d->c_out) * 16;
}

extern "C" int dn_spconv_pack_weights(const dn_conv_desc* d, const float* weight_oihw, float wmul,
                                      void* packed, void* stream) {
  if (int rc = validate(d)) return rc;
  DN_REQUIRE(weight_oihw && packed, "spconv pack: null pointer");
  DN_REQUIRE(d->c1 == 0 || d->c0 % 16 == 0, "spconv pack: concat needs c0 %% 16 == 0");
  const int taps = d->ksize * d->ksize, cp = cout_pad_of(d->c_out), nch = packed_chunks(*d);
  if (up_mode(*d) == 2)
    return dn::spq_pack_weights(weight_oihw, packed, d->c_out, d->c0 + d->c1, chunks_of(d->c0), cp, nch, wmul,
                                (hipStream_t)stream);
  if (up_merged(*d)) {
    hipLaunchKernelGGL(sp_pack_weights_up_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, weight_oihw,
                       (unsigned char*)packed, d->c_out, d->c0 + d->c1, chunks_of(d->c0), cp, nch, wmul);
    return dn::check_launch("sp_pack_weights_up_kernel");
  }
  const long total = (long)nch * taps * 2 * cp;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(sp_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     weight_oihw, (unsigned char*)packed, d->c_out, d->c0 + d->c1, taps, cp, nch, wmul,
                     total);
  return dn::check_launch("sp_pack_weights_kernel");
}

extern "C" size_t dn_spconv_pack_multi_table_bytes(int n_jobs) { return n_jobs > 0 ? sizeof(PackJob) * (size_t)n_jobs : 0; }

extern "C" int dn_spconv_pack_multi_prepare(const dn_pack_job* jobs, int n_jobs, void* table_host, int* total_blocks) {
  DN_REQUIRE(jobs && table_host && total_blocks && n_jobs > 0, "spconv pack multi: null pointer / no jobs");
  PackJob* t = static_cast<PackJob*>(table_host);
  int first = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const dn_pack_job& q = jobs[i];
    const dn_conv_desc* d = &q.desc;
    if (int rc = validate(d)) return rc;
    DN_REQUIRE(q.weight && q.packed, "spconv pack multi: job %d: null pointer", i);
    DN_REQUIRE(d->c1 == 0 || d->c0 % 16 == 0, "spconv pack multi: job %d: concat needs c0 %% 16 == 0", i);
    if (up_mode(*d) != 0)
      return dn::fail(DN_ERR_UNSUPPORTED, "spconv pack multi: job %d is packed tap-merged (dn_spconv_pack_weights only)", i);
    const int c_in = d->c0 + d->c1, taps = d->ksize * d->ksize;
    DN_REQUIRE(q.mode >= 0 && q.mode <= 2, "spconv pack multi: job %d: mode %d", i, q.mode);
    if (q.mode == 0)
      DN_REQUIRE(q.ci_first >= 0 && q.ci_first + c_in <= q.cin_total, "spconv pack multi: job %d: columns %d + %d of %d", i,
                 q.ci_first, c_in, q.cin_total);
    if (q.mode == 1)
      DN_REQUIRE(q.ci_first >= 0 && q.ci_first + d->c_out <= q.cin_total, "spconv pack multi: job %d: columns %d + %d of %d", i,
                 q.ci_first, d->c_out, q.cin_total);
    if (q.mode == 2)
      DN_REQUIRE(taps == 9 && q.n_in > 0 && d->c_out == 4 * q.n_in && q.ci_first >= 0 && q.ci_first + q.n_in <= q.cin_total,
                 "spconv pack multi: job %d: class form needs a 3x3 layer of 4 * n_in outputs", i);
    PackJob& j = t[i];
    j.w = q.weight;
    j.out = static_cast<unsigned char*>(q.packed);
    j.c_out = d->c_out; j.c_in = c_in; j.taps = taps; j.cout_pad = cout_pad_of(d->c_out); j.nchunks = packed_chunks(*d);
    j.mode = q.mode; j.cin_total = q.cin_total; j.ci_first = q.ci_first; j.n_in = q.n_in;
    j.wmul = q.wmul; j.pad_ = 0; j.padf_ = 0.f;
    j.total = (long)j.nchunks * taps * 2 * j.cout_pad;
    j.n_blocks = (int)((j.total + 255) / 256 < 4096 ? (j.total + 255) / 256 : 4096);      // the single launch's grid
    j.block_first = first;
    first += j.n_blocks;
  }
  *total_blocks = first;
  return DN_OK;
}

extern "C" int dn_spconv_pack_weights_multi(const void* table_device, int n_jobs, int total_blocks, void* stream) {
  DN_REQUIRE(table_device && n_jobs > 0 && total_blocks > 0, "spconv pack multi: bad arguments");
  hipLaunchKernelGGL(sp_pack_weights_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const PackJob*>(table_device), n_jobs);
  return dn::check_launch("sp_pack_weights_multi_kernel");
}

extern "C" size_t dn_sp_post1x1_packed_bytes(void) { return 2 * 4 * 2 * 2 * 32 * 16; }

extern "C" int dn_sp_post1x1_pack_weights(const float* w2, int c_out2, int c_in2, float wmul,
                                          void* packed, void* stream) {
  DN_REQUIRE(w2 && packed, "sp post1x1 pack: null pointer");
  DN_REQUIRE(c_out2 > 0 && c_out2 <= 64 && c_in2 > 0 && c_in2 <= 64,
             "sp post1x1 pack: c_out2 %d / c_in2 %d must be in 1..64", c_out2, c_in2);
  hipLaunchKernelGGL(sp_pack_post_kernel, dim3(2), dim3(256), 0, (hipStream_t)stream, w2,
                     (unsigned char*)packed, c_out2, c_in2, wmul);
  return dn::check_launch("sp_pack_post_kernel");
}

namespace {
// w2 [c_out2][64] block-diagonal (rows < split read columns 0..31, the rest columns 32..63)
// -> [cb 2][nt 2][ks 2][part 2][h 2][n 32] pieces
__global__ void sp_pack_heads_kernel(const float* __restrict__ w2, unsigned char* __restrict__ out,
                                     int c_out2, int split, float wmul) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // (cb, nt, ks, h, n)
  if (idx >= 2 * 2 * 2 * 2 * 32) return;
  const int n = idx % 32, h = (idx / 32) % 2, ks = (idx / 64) % 2, nt = (idx / 128) % 2, cb = idx / 256;
  const int rows = cb ? c_out2 - split : split, row0 = cb ? split : 0;
  half8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int r = nt * 32 + n, k = ks * 16 + h * 8 + e;
    float v = r < rows ? w2[(size_t)(row0 + r) * 64 + cb * 32 + k] * wmul : 0.f;
    v = fminf(fmaxf(v, -65504.f), 65504.f);
    hi[e] = (_Float16)v;
    lo[e] = (_Float16)(v - (float)hi[e]);
  }
  unsigned char* o = out + (size_t)((((((cb * 2 + nt) * 2 + ks) * 2 + 0) * 2 + h) * 32 + n)) * 16;
  *reinterpret_cast<half8*>(o) = hi;
  *reinterpret_cast<half8*>(o + 2 * 32 * 16) = lo;
}
}  // namespace

extern "C" int dn_sp_post1x1_pack_heads(const float* w2, int c_out2, int split, float wmul, void* packed,
                                        void* stream) {
  DN_REQUIRE(w2 && packed, "sp heads pack: null pointer");
  DN_REQUIRE(split > 0 && split < c_out2 && split <= 64 && c_out2 - split <= 64,
             "sp heads pack: split %d of %d outputs", split, c_out2);
  hipLaunchKernelGGL(sp_pack_heads_kernel, dim3(2), dim3(256), 0, (hipStream_t)stream, w2,
                     (unsigned char*)packed, c_out2, split, wmul);
  return dn::check_launch("sp_pack_heads_kernel");
}

extern "C" int dn_spconv_force_config(int cfg) {
  g_sp_force = cfg;
  return DN_OK;
}

extern "C" int dn_spconv_set_upmode(int mode) {
  g_sp_upmode = mode;
  return DN_OK;
}

namespace {
int spconv2d_impl(const dn_conv_desc* d, const void* src0, const void* src1, const void* packed, const float* scale,
                  const float* shift, void* out, float* out_nhwc, int ld_nhwc, void* stream, int kslices = 1,
                  void* workspace = nullptr, size_t workspace_bytes = 0);
}

extern "C" int dn_spconv2d(const dn_conv_desc* d, const void* src0, const void* src1,
                           const void* packed, const float* scale, const float* shift, void* out,
                           void* stream) {
  return spconv2d_impl(d, src0, src1, packed, scale, shift, out, nullptr, 0, stream);
}

extern "C" int dn_spconv2d_dual(const dn_conv_desc* d, const void* src0, const void* src1, const void* packed,
                                const float* scale, const float* shift, void* out_sp, float* out_nhwc, int ld_nhwc,
                                void* stream) {
  DN_REQUIRE(d && out_nhwc && ld_nhwc >= d->c_out && ld_nhwc % 4 == 0 && d->c_out % 4 == 0 &&
                 (reinterpret_cast<uintptr_t>(out_nhwc) & 15) == 0,
             "spconv dual: the fp32 NHWC output needs c_out %% 4 == 0, ld >= c_out, ld %% 4 == 0, 16-byte alignment");
  return spconv2d_impl(d, src0, src1, packed, scale, shift, out_sp, out_nhwc, ld_nhwc, stream);
}

extern "C" int dn_spconv2d_nhwc(const dn_conv_desc* d, const void* src0, const void* src1, const void* packed,
                                const float* scale, const float* shift, float* out_nhwc, int ld_nhwc, void* stream) {
  DN_REQUIRE(d && out_nhwc && ld_nhwc >= d->c_out && ld_nhwc % 4 == 0 && d->c_out % 4 == 0 &&
                 (reinterpret_cast<uintptr_t>(out_nhwc) & 15) == 0,
             "spconv nhwc: the fp32 NHWC output needs c_out %% 4 == 0, ld >= c_out, ld %% 4 == 0, 16-byte alignment");
  return spconv2d_impl(d, src0, src1, packed, scale, shift, nullptr, out_nhwc, ld_nhwc, stream);
}

// K-sliced form: does the layer qualify?  (3x3, no row-merged image, no hi-only source)
inline bool ks_layer(const dn_conv_desc& d) { return d.ksize == 3 && d.math != 3 && d.math != 4 && up_mode(d) != 1; }

// Can the layer run with `kslices` canonical K slices?  A property of the layer and of the process's up-conv form
// (DN_SP_UPMERGE / dn_spconv_set_upmode), never of the batch: callers that take the count from a table fall back to 1
// where this says no, instead of meeting DN_ERR_ARG in dn_spconv2d_ks.
extern "C" int dn_spconv_ks_supported(const dn_conv_desc* d, int kslices) {
  if (validate(d) != DN_OK) return 0;
  if (kslices == 1) return 1;
  if (kslices != 2 && kslices != 4) return 0;
  return ks_layer(*d) && chunks_of(d->c0) + chunks_of(d->c1) >= kslices ? 1 : 0;
}

extern "C" size_t dn_spconv_workspace_bytes(const dn_conv_desc* d, int kslices) {
  if (validate(d) != DN_OK || kslices <= 1 || !ks_layer(*d)) return 0;
  const size_t ho = out_dim(d->h_in, d->ksize, d->stride), wo = out_dim(d->w_in, d->ksize, d->stride);
  // every tile of the launch split, at the padding of the smallest tiles (8 rows, 32 columns, 64 channels)
  return (size_t)kslices * d->n_images * ((ho + 7) / 8 * 8) * ((wo + 31) / 32 * 32) * ((d->c_out + 63) / 64 * 64) * sizeof(float);
}

extern "C" int dn_spconv2d_ks(const dn_conv_desc* d, int kslices, const void* src0, const void* src1, const void* packed,
                              const float* scale, const float* shift, void* out, float* out_nhwc, int ld_nhwc,
                              void* workspace, size_t workspace_bytes, void* stream) {
  DN_REQUIRE(kslices == 1 || kslices == 2 || kslices == 4, "spconv: kslices must be 1, 2 or 4 (got %d)", kslices);
  DN_REQUIRE(workspace == nullptr || (reinterpret_cast<uintptr_t>(workspace) & 15) == 0, "spconv: workspace must be 16-byte aligned");
  DN_REQUIRE(!out_nhwc || (d && ld_nhwc >= d->c_out && ld_nhwc % 4 == 0 && d->c_out % 4 == 0 &&
                           (reinterpret_cast<uintptr_t>(out_nhwc) & 15) == 0),
             "spconv: the fp32 NHWC output needs c_out %% 4 == 0, ld >= c_out, ld %% 4 == 0, 16-byte alignment");
  return spconv2d_impl(d, src0, src1, packed, scale, shift, out, out_nhwc, ld_nhwc, stream, kslices, workspace, workspace_bytes);
}

namespace {
int spconv2d_impl(const dn_conv_desc* d, const void* src0, const void* src1, const void* packed, const float* scale,
                  const float* shift, void* out, float* out_nhwc, int ld_nhwc, void* stream, int kslices,
                  void* workspace, size_t workspace_bytes) {
  if (int rc = validate(d)) return rc;
  DN_REQUIRE(src0 && packed && scale && shift && (out || out_nhwc), "spconv: null pointer");
  DN_REQUIRE(d->c1 == 0 || src1, "spconv: c1 > 0 but src1 is null");
  SpArgs a;
  if (int rc = fill_args(d, src0, src1, packed, scale, shift, out, a)) return rc;
  a.out_b = out_nhwc; a.ldo_b = ld_nhwc;      // POST 0: optional fp32 NHWC copy of the output
  DN_REQUIRE(!out_nhwc || g_sp_force < 100, "spconv dual: not with a forced tools configuration");
  hipStream_t s = (hipStream_t)stream;
  // K slices: a property of the LAYER (the caller passes the same count whatever the batch): results do not depend
  // on how a launch distributes the slices.  Layers with fewer chunks than slices, 1x1 layers, the row-merged image
  // and hi-only sources are refused rather than silently computed in another order.
  DN_REQUIRE(kslices == 1 || (ks_layer(*d) && a.c0g + a.c1g >= kslices),
             "spconv: %d K slices need a 3x3 layer of at least that many 16-channel chunks", kslices);
  if (up_mode(*d) == 2) {   // the packed image is the quad-merged one: conv_spq.hip (tools: 20 / 21 force BN = 32 / 64)
    DN_REQUIRE(a.c1g == 0 || d->c0 % 16 == 0, "spconv: concat needs c0 %% 16 == 0");
    return dn::spq_conv(d, src0, src1, packed, (size_t)a.wpk_bytes, scale, shift, out, a.cout_pad,
                        g_sp_force == 20 ? 32 : g_sp_force == 21 ? 64 : g_sp_force == 22 ? 33 :
                        (g_sp_force >= 23 && g_sp_force <= 25) ? 78 + g_sp_force : 0, s, kslices, (float*)workspace,
                        workspace_bytes, out_nhwc, ld_nhwc);
  }
  if (kslices > 1) {
    a.ks.count = kslices;
    a.ks.partial = (float*)workspace;
    a.ks_ws_bytes = workspace_bytes;
    const SpCfg c = select_cfg(*d, kslices, workspace != nullptr && workspace_bytes > 0);
    switch (c.id) {
      //                                   KS S  TH TW  BN TG CA WM WN WTM WTN          KSL
      case S3_256x32:      return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 0, 0, 0, 0, 1>(a, *d, s);
      case S3_128x64:      return launch<3, 1, 8, 16, 64, 3, 1, 2, 2, 2, 1, 0, 0, 0, 0, 0, 1>(a, *d, s);
      case S3_64x64:       return launch<3, 1, 8, 8, 64, 3, 1, 2, 2, 1, 1, 0, 0, 0, 0, 0, 1>(a, *d, s);
      case S3S2_128x64:    return launch<3, 2, 8, 16, 64, 3, 1, 2, 2, 2, 1, 0, 0, 0, 0, 0, 1>(a, *d, s);
      case S3S2_64x64:     return launch<3, 2, 8, 8, 64, 3, 1, 2, 2, 1, 1, 0, 0, 0, 0, 0, 1>(a, *d, s);
      case S3_64x64_T9:    return launch<3, 1, 8, 8, 64, 9, 1, 2, 2, 1, 1, 0, 0, 0, 0, 0, 1>(a, *d, s);
      case S3_128x64_T9:   return launch<3, 1, 8, 16, 64, 9, 1, 2, 2, 2, 1, 0, 0, 0, 0, 0, 1>(a, *d, s);
      case S3S2_64x64_T9:  return launch<3, 2, 8, 8, 64, 9, 1, 2, 2, 1, 1, 0, 0, 0, 0, 0, 1>(a, *d, s);
      case S3S2_128x64_T9: return launch<3, 2, 8, 16, 64, 9, 1, 2, 2, 2, 1, 0, 0, 0, 0, 0, 1>(a, *d, s);
      default: return dn::fail(DN_ERR_UNSUPPORTED, "spconv: tile configuration %d has no K-sliced form", (int)c.id);
    }
  }
  if (d->math == 3) {   // hi-only source 0: the 8 x 32 x 32 tile, weight-stationary when the layer fits
    using T32 = SpTile<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 1, 0, 1>;
    if (fits_stationary(*d, 32, T32::A_STAGE, 0, 2)) return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 0, 1, 0, 1>(a, *d, s);
    return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 0, 0, 0, 1>(a, *d, s);
  }
  if (d->math == 4) {   // bit-grid source 0: the same tile, staged by expansion; weight-stationary only
    using T32 = SpTile<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 1, 0, 2>;
    if (fits_stationary(*d, 32, T32::A_STAGE, 0, 2)) return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 0, 1, 0, 2>(a, *d, s);
    return dn::fail(DN_ERR_UNSUPPORTED, "spconv: a bit-grid source needs a layer whose weights fit the LDS (c_out <= 32)");
  }
  const SpCfg c = select_cfg(*d);
  if (up_merged(*d)) {   // the packed image is the row-merged one: only the tiles that implement it
    switch (c.id) {
      case S3_256x64: return launch<3, 1, 8, 32, 64, 3, 1, 4, 1, 2, 2, 0, 0, 0, 1>(a, *d, s);
      case S3_256x32: return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 0, 0, 1>(a, *d, s);
      case S3_128x64: return launch<3, 1, 8, 16, 64, 3, 1, 2, 2, 2, 1, 0, 0, 0, 1>(a, *d, s);
      default: return dn::fail(DN_ERR_UNSUPPORTED, "spconv: tile configuration %d has no row-merged form", (int)c.id);
    }
  }
  if (g_sp_force >= 100 && d->ksize == 3 && d->stride == 1) {   // tools: timing-only ablations
    switch (g_sp_force) {
      case 101: return launch<3, 1, 8, 32, 64, 3, 1, 4, 1, 2, 2, 0, 1>(a, *d, s);
      case 102: return launch<3, 1, 8, 32, 64, 3, 1, 4, 1, 2, 2, 0, 2>(a, *d, s);
      case 103: return launch<3, 1, 8, 32, 64, 3, 1, 4, 1, 2, 2, 0, 3>(a, *d, s);
      case 104: return launch<3, 1, 8, 32, 64, 3, 1, 4, 1, 2, 2, 0, 4>(a, *d, s);
      case 105: return launch<3, 1, 8, 32, 64, 3, 1, 4, 1, 2, 2, 0, 5>(a, *d, s);
      case 201: return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 1>(a, *d, s);
      case 202: return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 2>(a, *d, s);
      case 203: return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 3>(a, *d, s);
      case 204: return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 4>(a, *d, s);
      case 205: return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 5>(a, *d, s);
      case 206: return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 6>(a, *d, s);
      case 207: return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 7>(a, *d, s);
      case 301: return launch<3, 1, 8, 8, 64, 9, 1, 2, 2, 1, 1, 0, 1>(a, *d, s);   // the deep-regime tile
      case 302: return launch<3, 1, 8, 8, 64, 9, 1, 2, 2, 1, 1, 0, 2>(a, *d, s);
      case 303: return launch<3, 1, 8, 8, 64, 9, 1, 2, 2, 1, 1, 0, 3>(a, *d, s);
      case 304: return launch<3, 1, 8, 8, 64, 9, 1, 2, 2, 1, 1, 0, 4>(a, *d, s);
      case 305: return launch<3, 1, 8, 8, 64, 9, 1, 2, 2, 1, 1, 0, 5>(a, *d, s);
      default: break;
    }
  }
#if DN_S2_ABL   // tools/ab build only (AB_FILES=conv_sp tools/ab/build.sh DN_S2_ABL 1): timing-only ablations of the stride-2 8 x 8 tile
  if (g_sp_force >= 400 && d->ksize == 3 && d->stride == 2) {
    switch (g_sp_force) {
      case 400: return launch<3, 2, 8, 8, 64, 3, 1, 2, 2, 1, 1, 0, 0>(a, *d, s);   // as shipped in round 4 (two weight stages)
      case 401: return launch<3, 2, 8, 8, 64, 3, 1, 2, 2, 1, 1, 0, 1>(a, *d, s);   // no weight DMA after the first step
      case 402: return launch<3, 2, 8, 8, 64, 3, 1, 2, 2, 1, 1, 0, 2>(a, *d, s);   // no patch DMA after the first chunk
      case 403: return launch<3, 2, 8, 8, 64, 3, 1, 2, 2, 1, 1, 0, 3>(a, *d, s);   // neither
      case 404: return launch<3, 2, 8, 8, 64, 3, 1, 2, 2, 1, 1, 0, 4>(a, *d, s);   // no epilogue stores
      case 405: return launch<3, 2, 8, 8, 64, 3, 1, 2, 2, 1, 1, 0, 5>(a, *d, s);   // 3 + operands from registers: the MFMA stream alone
      case 407: return launch<3, 2, 8, 8, 64, 3, 1, 2, 2, 1, 1, 0, 7>(a, *d, s);   // 5 + no stores, no barriers
      default: break;
    }
  }
#endif
  // weight-stationary forms (short-K full-resolution layers): two workgroups per CU
  // DN_SP_STATIONARY=0: weights streamed per step everywhere (A/B runs)
  static const int stat_env = [] { const char* e = getenv("DN_SP_STATIONARY"); return e ? atoi(e) : 1; }();
  // (a 16x32-pixel stationary tile -- four MFMA tiles per wave, ONE workgroup per CU -- measured
  // 14 % slower on conv8_2 and 33 % slower on the heads than 8x32 with two workgroups per CU, and
  // weights held in registers instead of LDS spilled at the 256-VGPR budget of two workgroups: what
  // these short-K layers need is a second workgroup to run under the first one's waits)
  if (stat_env && g_sp_force < 0 && c.id == S3_256x32) {
    using T32 = SpTile<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 1>;
    if (fits_stationary(*d, 32, T32::A_STAGE, 0, 2))
      return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 0, 1>(a, *d, s);
  }
  if (g_sp_force == S3_256x32_ST) return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 0, 0, 1>(a, *d, s);
  // Three weight stages (round 5) on the 8 x 8-pixel tiles, whose steps (9 MFMAs per wave) are far shorter than the loaded L2 latency
  // and whose LDS has the room (stride 2: 78 KB, two workgroups per CU as before; stride 1: 53 KB, three as before).  DN_SP_B3: bit 0 =
  // the stride-2 tile, bit 1 = the stride-1 tile (A/B runs; the results are bit-identical -- same operands, same MFMA order).
  static const int b3_env = [] { const char* e = getenv("DN_SP_B3"); return e ? atoi(e) : 3; }();
  if (g_sp_force < 0 && c.id == S3S2_64x64 && (b3_env & 1))
    return launch<3, 2, 8, 8, 64, 3, 1, 2, 2, 1, 1, 0, 0, 0, 0, 0, 0, 3>(a, *d, s);
  if (g_sp_force < 0 && c.id == S3_64x64 && (b3_env & 2))
    return launch<3, 1, 8, 8, 64, 3, 1, 2, 2, 1, 1, 0, 0, 0, 0, 0, 0, 3>(a, *d, s);
  switch (c.id) {
    //                               KS S  TH TW  BN TG CA WM WN WTM WTN
    case S3_256x64:   return launch<3, 1, 8, 32, 64, 3, 1, 4, 1, 2, 2>(a, *d, s);
    case S3_256x32:   return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1>(a, *d, s);
    case S3_128x64:   return launch<3, 1, 8, 16, 64, 3, 1, 2, 2, 2, 1>(a, *d, s);
    case S3_64x64:    return launch<3, 1, 8, 8, 64, 3, 1, 2, 2, 1, 1>(a, *d, s);
    case S3S2_128x64: return launch<3, 2, 8, 16, 64, 3, 1, 2, 2, 2, 1>(a, *d, s);
    case S3S2_64x64:  return launch<3, 2, 8, 8, 64, 3, 1, 2, 2, 1, 1>(a, *d, s);
    case S1_256x64:   return launch<1, 1, 8, 32, 64, 1, 2, 4, 1, 2, 2>(a, *d, s);
    case S1_64x64:    return launch<1, 1, 8, 8, 64, 1, 4, 2, 2, 1, 1>(a, *d, s);
    case S3_256x64_T9: return launch<3, 1, 8, 32, 64, 9, 1, 4, 1, 2, 2>(a, *d, s);
    case S3_512x64:   return launch<3, 1, 16, 32, 64, 3, 1, 8, 1, 2, 2>(a, *d, s);
    case S3_256x128:  return launch<3, 1, 8, 32, 128, 3, 1, 4, 2, 2, 2>(a, *d, s);
    case S1_256x64_C1: return launch<1, 1, 8, 32, 64, 1, 1, 4, 1, 2, 2>(a, *d, s);
    case S3_64x64_T9:  return launch<3, 1, 8, 8, 64, 9, 1, 2, 2, 1, 1>(a, *d, s);
    case S3_128x64_T9: return launch<3, 1, 8, 16, 64, 9, 1, 2, 2, 2, 1>(a, *d, s);
    case S3S2_64x64_T9: return launch<3, 2, 8, 8, 64, 9, 1, 2, 2, 1, 1>(a, *d, s);
    case S3S2_128x64_T9: return launch<3, 2, 8, 16, 64, 9, 1, 2, 2, 2, 1>(a, *d, s);
    default: break;
  }
  return dn::fail(DN_ERR_UNSUPPORTED, "spconv: no tile configuration");
}
}  // namespace

extern "C" int dn_spconv2d_post1x1(const dn_conv_desc* d, const dn_post1x1_desc* p, const void* src0,
                                   const void* src1, const void* packed, const float* scale,
                                   const float* shift, const void* packed2, const float* scale2,
                                   const float* shift2, int out_f32, void* out_a, float* out_b,
                                   void* stream) {
  if (int rc = validate(d)) return rc;
  DN_REQUIRE(p && src0 && packed && scale && shift && packed2 && scale2 && shift2 && out_a,
             "spconv+1x1: null pointer");
  DN_REQUIRE(d->ksize == 3 && d->stride == 1 && d->c_out == 64,
             "spconv+1x1: needs a 3x3 stride-1 conv with 64 output channels");
  DN_REQUIRE(p->c_out2 > 0 && p->c_out2 <= 64, "spconv+1x1: c_out2 %d must be in 1..64", p->c_out2);
  if (out_f32) {
    DN_REQUIRE(p->c_out2 % 4 == 0 && p->split % 4 == 0 && p->split > 0 && p->split <= p->c_out2,
               "spconv+1x1: c_out2 %d / split %d must be multiples of 4, split in (0, c_out2]",
               p->c_out2, p->split);
    DN_REQUIRE(p->ldo_a >= p->split && p->ldo_a % 4 == 0, "spconv+1x1: bad ldo_a");
    DN_REQUIRE(p->split == p->c_out2 || (out_b && p->ldo_b >= p->c_out2 - p->split && p->ldo_b % 4 == 0),
               "spconv+1x1: second output missing or too narrow");
    DN_REQUIRE((reinterpret_cast<uintptr_t>(out_b) & 15) == 0, "spconv+1x1: out_b must be 16-byte aligned");
  }
  SpArgs a;
  if (int rc = fill_args(d, src0, src1, packed, scale, shift, out_a, a)) return rc;
  DN_REQUIRE((reinterpret_cast<uintptr_t>(packed2) & 15) == 0, "spconv+1x1: packed2 must be 16-byte aligned");
  a.w2 = (const unsigned char*)packed2; a.scale2 = scale2; a.shift2 = shift2; a.out_b = out_b;
  a.c_out2 = p->c_out2; a.relu2 = p->relu2; a.split2 = p->split; a.ldo_a = p->ldo_a; a.ldo_b = p->ldo_b;
  a.post_f32 = out_f32 ? 1 : 0;
  a.cog = chunks_of(p->c_out2);   // SP output: the second stage's channels
  if (p->block_diag) {
    DN_REQUIRE(out_f32 && p->split < p->c_out2 && p->split <= 64 &&
                   p->c_out2 - p->split <= 64,
               "spconv+1x1: the block-diagonal form needs two fp32 outputs of <= 64 columns each");
    using TS = SpTile<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 2, 1>;
    static const int stat_env = [] { const char* e = getenv("DN_SP_STATIONARY"); return e ? atoi(e) : 1; }();
    // both heads in one workgroup (the input patch is staged once; -4 % on the heads launch): 1 = streaming
    // weights (default), 2 = LDS-resident weights, 0 = one head per workgroup
    static const int heads64 = [] { const char* e = getenv("DN_SP_HEADS64"); return e ? atoi(e) : 1; }();
    using TS64 = SpTile<3, 1, 8, 32, 64, 3, 1, 4, 1, 2, 2, 2, 1>;
    if (heads64 == 2 && fits_stationary(*d, 64, TS64::A_STAGE, 1024, 1, true))
      return launch<3, 1, 8, 32, 64, 3, 1, 4, 1, 2, 2, 2, 0, 1>(a, *d, (hipStream_t)stream);
    if (heads64 >= 1) return launch<3, 1, 8, 32, 64, 3, 1, 4, 1, 2, 2, 2, 0, 0>(a, *d, (hipStream_t)stream);
    // ablations of the heads launch (measurement only): 3 = no operand DMA, 4 = no stores, 5 = no LDS reads
    static const int heads_abl = [] { const char* e = getenv("DN_SP_HEADS_ABL"); return e ? atoi(e) : 0; }();
    if (heads_abl == 3) return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 2, 3, 0>(a, *d, (hipStream_t)stream);
    if (heads_abl == 4) return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 2, 4, 0>(a, *d, (hipStream_t)stream);
    if (heads_abl == 5) return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 2, 5, 0>(a, *d, (hipStream_t)stream);
    if (heads_abl == 9) return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 2, 0, 0>(a, *d, (hipStream_t)stream);
    if (stat_env && fits_stationary(*d, 32, TS::A_STAGE, 1024, 2, true))
      return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 2, 0, 1>(a, *d, (hipStream_t)stream);
    return launch<3, 1, 8, 32, 32, 3, 1, 4, 1, 2, 1, 2, 0, 0>(a, *d, (hipStream_t)stream);
  }
  {
    using TP = SpTile<3, 1, 8, 32, 64, 3, 1, 4, 1, 2, 2, 1, 1>;
    static const int stat_env = [] { const char* e = getenv("DN_SP_STATIONARY"); return e ? atoi(e) : 1; }();
    const int extra = TP::W2_BYTES + (out_f32 ? TP::NW * 32 * (p->c_out2 + 4) * 4 : 0) + 1024;   // + the static affine block
    if (stat_env && g_sp_force != 0 && fits_stationary(*d, 64, TP::A_STAGE, extra, 1))
      return launch<3, 1, 8, 32, 64, 3, 1, 4, 1, 2, 2, 1, 0, 1>(a, *d, (hipStream_t)stream);
  }
  return launch<3, 1, 8, 32, 64, 3, 1, 4, 1, 2, 2, 1>(a, *d, (hipStream_t)stream);
}

#include "conv_pre_pair.inl"
