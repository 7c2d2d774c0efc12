// K5 + K6 in one launch: the whole per-pixel pairwise attention MLP of the DiscoGraph fusion
// (2C -> 128 -> 32 -> 8 -> 1), the exp / sum softmax over the agents of a scene and the weighted
// sum of the (warped) neighbour maps -- lanes over PIXELS, every layer on the f16 MFMA (split-f16
// x3, fp32 accumulate, the conv engine's arithmetic), the activations of all four layers in
// registers from the first operand load to the last store.
//
// Replaces the chain  dn_conv2d(mlp_g) -> dn_conv2d(mlp_f) -> dn_disco_fuse_tail -> dn_sp_from_nhwc
// (g and fw, 63 MB per step, are never materialised; the maps are read where they lie).
//
// One WAVE owns 32 consecutive pixels of one (sample, ego).  Lane (j, h) holds, of pixel j, the
// channels {16 ks + 8 h + e}: exactly the B-operand fragment of v_mfma_f32_32x32x16_f16, so an
// NHWC row piece of 8 floats becomes an MFMA operand with one split and no LDS transpose.  Layer
// outputs come back as D[i = unit][j = pixel] with lane (j, h) holding units 8g + 4h + e; two
// v_permlane32_swap per register pair turn them into the next layer's B fragments (the trick of
// conv_sp.hip's fused 1x1 stage).  Weights are read as A-operand fragments straight from L2.
//
//   pass 1, per neighbour k (ego first, then j ascending, as the reference's list):
//     acc = W1_ego . x_ego (once)  +  W1_nbr . y_k ;  h1 = relu(bn1(acc + b1)) ; h2, h3 likewise;
//     s_k = relu(w4 . h3 + b4) ;  e_k = exp(s_k)                       (no max-shift, as upstream)
//   pass 2:  w_k = e_k / sum_k e_k ;  fused = sum_k w_k * y_k   (fp32, list order; y_k re-read: the
//     rows were read microseconds ago and sit in L2), written as split-planar pieces for the decoder
//     and/or as fp32 NHWC rows.
//
// Replaces PixelWeightedFusionSoftmax.forward and the fusion loop body of
// upstream:coperception/models/det/DiscoNet.py :: DiscoNet.forward (Appx A.5).
#include "dn_internal.h"
#include "sp_device.h"
#include <cstdlib>
#include <type_traits>

// tools/ab (DN_FUSE_PHASES 1): every active wave of the weight-in-LDS form adds its cycles per phase into g_fuse_phase --
// [0] staging + barriers, [1] ego term, [2] layer-1 passes of the slots, [3] tails, [4] weighted sum + stores, [5] waves, [6] total,
// [7] first cycle count seen (unused) -- read with dn_fuse_phase_cycles().  Never in the shipped build.
#ifndef DN_FUSE_PHASES
#define DN_FUSE_PHASES 0
#endif
#if DN_FUSE_PHASES
__device__ unsigned long long g_fuse_phase[8];
extern "C" int dn_fuse_phase_cycles(unsigned long long* host8, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (host8 && hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_fuse_phase), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_fuse_phase), z, sizeof z) != hipSuccess) return -1; }
  return 0;
}
#endif

namespace {

constexpr int MAX_AGENTS = 8;
constexpr size_t kW2Bytes = 8 * 2 * 2 * 32 * 16, kW3Bytes = 2 * 2 * 2 * 32 * 16;
constexpr int FUSE_G = 3;   // list slots per layer-1 pass: 3 accumulator sets (192 AGPRs) is what hipcc allocates without spilling
#ifndef DN_FUSE_RING
#define DN_FUSE_RING 4   // row ring of the weight-in-LDS form's layer-1 loop (k-steps); 8 measured the same (67.5 vs 67.9 us standalone): the loop does not wait for its rows
#endif
#ifndef DN_FUSE_GL
#define DN_FUSE_GL 2      // 3 spills 16 registers there (tools/kernel_resources.py fuse_mlp); the grouping does not change a bit of any slot's chain
#endif
constexpr int FUSE_GL = DN_FUSE_GL;   // the same for the weight-in-LDS form (the ego term holds 64 more registers there)

struct FuseMlpArgs {
  const float* feat;
  const float* warped;
  const int32_t* num_agent;
  const unsigned char* w1;   // [mat 2 (ego, nbr)][nt 4][ks C/16][part 2][h 2][row 32] x 16 B
  const unsigned char* w2;   // [ks 8][part 2][h 2][row 32] x 16 B   (32 x 128)
  const unsigned char* w3;   // [ks 2][part 2][h 2][row 32] x 16 B   (8 x 32, rows 8..31 zero)
  const float *s1, *t1, *s2, *t2, *s3, *t3, *w4, *b4;
  unsigned char* fused_sp;
  float* fused_nhwc;
  float* weights_out;
  int batch, agents, hw, only_v2i, ego_first, ego_count, tiles;
  int total_tiles;   // batch * ego_count * tiles (the weight-in-LDS form has workgroups of several tiles)
  int warped_fm;   // `warped` is fragment-major (dn_warp_neighbors_fm): a k-step of a tile is two contiguous 1 KB runs
};
// Timing-only ablations (results are garbage), COMPILE-time so that the shipped kernel carries none of it: build a variant
// with tools/ab/build.sh DN_FUSE_ABL <mask>: 1 = pass 2 reads no rows, 2 = layer 1 reads each row's first k-step only,
// 4 = no layers 2-4, 8 = layer 1 splits no operands.  (As a run-time argument the four tests cost 4 us of the launch.)
#ifndef DN_FUSE_ABL
#define DN_FUSE_ABL 0
#endif
constexpr int kAbl = DN_FUSE_ABL;
// layer-1 weight fragments (L2): how many k-steps ahead of their MFMAs they are requested (1: two register sets, shipped).  2 and 3
// (ring of four sets, no spills, counted vmcnt waits in the ISA) measured 83.7 us against 84.5 in the same lease: the launch is not
// waiting on the L2 latency of these loads.
#ifndef DN_FUSE_WAHEAD
#define DN_FUSE_WAHEAD 1
#endif
constexpr int kWAhead = DN_FUSE_WAHEAD, kWRing = kWAhead == 1 ? 2 : 4;
static_assert(kWAhead >= 1 && kWAhead <= 3, "weight prefetch distance");

__device__ inline half8 frag_of(const unsigned char* base, int idx) {
  return *reinterpret_cast<const half8*>(base + (size_t)idx * 16);
}

// G = neighbour-list slots processed together: the layer-1 weight fragments of a k-step are read
// once per group (they are the kernel's L2 traffic: 128 KB per pass at C = 256) and each slot adds
// 12 MFMAs behind them, which is what hides the fragment loads' latency.  One wave per workgroup:
// G accumulator sets of 64 registers need the whole 512-entry register file of a SIMD lane.
//
// NW = 4 (launches of fewer than 512 tiles): FOUR waves share the 32 pixels.  One wave per 32 pixels is 640 waves for 1024 SIMDs at
// the BASELINE size (128 for one rank's share of the agent-sharded step), each a serial chain of 6 layer-1 passes:
// the kernel ran at the latency of that chain.  With four waves the ego term E is computed one 32-unit tile per
// wave, the list slots go round-robin to the waves (slot k to wave k % 4, one slot per layer-1 pass), and
// pass 2 splits the channels (KS / 4 k-steps per wave).  Every output is produced by the same instruction
// sequence on the same operands as with NW = 1: the results are bit-identical (tests/test_gpu_fusion.py).
// WL > 0 (round 5): the WEIGHT-IN-LDS form for launches of at least two tiles per CU (the BASELINE size: 640 tiles).  A workgroup
// is WL waves, each the whole chain of ITS OWN tile (NW = 1 arithmetic, instruction for instruction), and the layer-1 weights --
// first W1_ego, then W1_nbr, 128 KB each at C = 256 -- are staged ONCE per workgroup into LDS, from where every k-step's eight
// fragments come at LDS latency.  In the one-wave form every wave streamed those fragments from L2 for itself (128 KB per layer-1
// pass, 245 MB per step) with nothing to switch to while a load was in flight: two thirds of the launch was that loop at 30 % MFMA
// utilisation.  The ego term stays in registers (no e_s), so the static LDS beside the 128 KB is 28 KB.
template <int C, int G, int NW, int WL = 0>
__global__ void __launch_bounds__(64 * (WL > 0 ? WL : NW), (NW == 1 || WL > 0) ? 1 : 2) disco_fuse_mlp_kernel(const FuseMlpArgs a) {
  constexpr bool kWL = WL > 0;
  constexpr int TWV = kWL ? WL : 1;        // waves of the workgroup that own a tile each
  static_assert(!kWL || NW == 1, "weight-in-LDS form: one wave per tile");
  constexpr int KS = C / 16;
  constexpr int KSW = KS / NW;             // k-steps (16-channel chunks) of pass 2 / the pass-through per wave
  static_assert(KS % NW == 0, "pass 2 splits the k-steps over the waves");
  constexpr int kW1MatBytes = 4 * KS * 2 * 2 * 32 * 16;      // one matrix (ego or nbr) of layer 1 as fragments
  extern __shared__ __attribute__((aligned(16))) unsigned char w1_s[];   // kWL: kW1MatBytes, the matrix in use
  __shared__ float ek_s[TWV][MAX_AGENTS][64];   // exp(s_k) of a wave's pixels, per list slot
  __shared__ float e_s[kWL ? 1 : 64][64];       // layer-1 ego term of the tile's pixels: [reg][lane] (kWL: in registers)
  __shared__ int jl_s[TWV][MAX_AGENTS];         // agent of each list slot (slot 0 = the ego)
  // layers 2-4: weight fragments and affines, read by every tail() -- LDS-resident (an L2 round
  // trip per dependent stage of the tail was ~40 % of the kernel)
  __shared__ __attribute__((aligned(16))) unsigned char w23_s[kW2Bytes + kW3Bytes];
  __shared__ __attribute__((aligned(16))) float aff_s[2 * 128 + 2 * 32 + 3 * 8];   // s1 t1 s2 t2 s3 t3 w4

  const int lane = threadIdx.x & 63, wave = NW > 1 ? threadIdx.x >> 6 : 0;
  const int tw = kWL ? threadIdx.x >> 6 : 0;   // which of the workgroup's tiles this wave owns
  const int ks_first = wave * KSW;
  const int li = lane & 31, lh = lane >> 5;
  const int wid_raw = kWL ? blockIdx.x * WL + tw : blockIdx.x;
  const bool have = wid_raw < a.total_tiles;        // kWL: the last workgroup may hold waves without a tile
  const int wid = have ? wid_raw : a.total_tiles - 1;
  const int tile = wid % a.tiles, il = (wid / a.tiles) % a.ego_count, b = wid / (a.tiles * a.ego_count);
  const int i = a.ego_first + il;
  const int p = tile * 32 + li;
  const bool pvalid = p < a.hw && have;
  const int pc = p < a.hw ? p : a.hw - 1;
  int live = a.num_agent[b];
  live = live < 0 ? 0 : (live < a.agents ? live : a.agents);   // never index past the agents that exist
  const size_t oimg = (size_t)il * a.batch + b;

  // A row = this lane's view of one map: piece (ks, r) = 4 floats at p + ks * kss + r * r1.  NHWC rows (the ego map,
  // `warped` in pixel-major form): kss = 16, r1 = 4; fragment-major `warped`: this lane's slot of the tile's k-step
  // block, kss = 512, r1 = 256 -- every load instruction of the wave is then one contiguous 1 KB run.
  struct Row {
    const float* p;
    int kss, r1;
  };
  const Row xrow_r = {a.feat + (((size_t)i * a.batch + b) * a.hw + pc) * C + 8 * lh, 16, 4};
  const float* xrow = xrow_r.p;
  auto yrow_of = [&](int j) {
    const int jj = j - (j > i ? 1 : 0);
    const float* pair = a.warped + (((size_t)b * a.ego_count + il) * (a.agents - 1) + jj) * a.hw * C;
    return a.warped_fm ? Row{pair + ((size_t)tile * KS * 2 * 64 + lane) * 4, 512, 256} : Row{pair + (size_t)pc * C + 8 * lh, 16, 4};
  };

  float amax = 0.f;   // max |value| split into the SP output (range flags, sp_device.h)
  bool nan_seen = false;
  // ---- store helpers: this lane's 8 channels of k-step ks
  auto store_piece = [&](int ks, const f32x4 v0, const f32x4 v1) {
    if (!pvalid) return;
    if (a.fused_sp) {
      u32x2 h0, l0, h1, l1;
      note_nan4(nan_seen, v0);
      note_nan4(nan_seen, v1);
      split4(v0, h0, l0, amax);
      split4(v1, h1, l1, amax);
      const size_t plane = (size_t)a.hw * 16;
      unsigned char* o = a.fused_sp + ((oimg * KS + ks) * 4 + lh) * plane + (size_t)p * 16;
      *reinterpret_cast<u32x4*>(o) = u32x4{h0[0], h0[1], h1[0], h1[1]};
      *reinterpret_cast<u32x4*>(o + 2 * plane) = u32x4{l0[0], l0[1], l1[0], l1[1]};
    }
    if (a.fused_nhwc) {
      float* o = a.fused_nhwc + (oimg * a.hw + p) * C + 16 * ks + 8 * lh;
      *reinterpret_cast<f32x4*>(o) = v0;
      *reinterpret_cast<f32x4*>(o + 4) = v1;
    }
  };

  constexpr int NTHR = 64 * (kWL ? WL : NW);
#if DN_FUSE_PHASES
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = __builtin_readcyclecounter();
  const unsigned long long t_begin = t_prev;
  auto mark = [&](int k) { const unsigned long long t = __builtin_readcyclecounter(); ph[k] += t - t_prev; t_prev = t; };
#else
  auto mark = [&](int) {};
#endif
  if (kWL || i < live) {      // (kWL: the waves of a workgroup own different egos -- everyone stages)
    const int t = threadIdx.x;
    for (int q = t; q < (int)((kW2Bytes + kW3Bytes) / 16); q += NTHR)   // w3 follows w2 in the packed block
      *reinterpret_cast<u32x4*>(w23_s + q * 16) = *reinterpret_cast<const u32x4*>(a.w2 + (size_t)q * 16);
    for (int q = t; q < 128; q += NTHR) { aff_s[q] = a.s1[q]; aff_s[128 + q] = a.t1[q]; }
    if (t < 32) { aff_s[256 + t] = a.s2[t]; aff_s[288 + t] = a.t2[t]; }
    if (t < 8) { aff_s[320 + t] = a.s3[t]; aff_s[328 + t] = a.t3[t]; aff_s[336 + t] = a.w4[t]; }
  }
  // kWL: one matrix of layer 1 -> LDS, a linear copy (fragment (nt, ks, part) of lane l keeps its place) by LDS-DMA: every wave
  // issues its share of the 1 KB instructions back to back, so the whole matrix is in flight at once -- one memory latency per
  // matrix.  (Through registers, 8 x 16 B per thread in flight, the two matrices cost ~25 us of the launch: the copy ran at the
  // latency of its 11 round trips.)
  auto stage_w1 = [&](int mat) {
    if constexpr (kWL) {
      const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.w1 + (size_t)mat * kW1MatBytes), 0,
                                                        kW1MatBytes, 0x00020000);
      constexpr int INSTR = kW1MatBytes / 1024;
      for (int q = tw; q < INSTR; q += WL) dma16(rs, w1_s + q * 1024, (unsigned)(q * 1024 + lane * 16), 0);
      wait_vm0();
    }
  };
  stage_w1(0);
  const float b4v = a.b4[0];   // read once: a global load inside every tail() sat on its critical path
  const unsigned char* w2l = w23_s;
  const unsigned char* w3l = w23_s + kW2Bytes;
  const float *s1l = aff_s, *t1l = aff_s + 128, *s2l = aff_s + 256, *t2l = aff_s + 288, *s3l = aff_s + 320,
              *t3l = aff_s + 328, *w4l = aff_s + 336;

  const bool active = have && i < live;      // kWL: a wave without a live ego still meets the workgroup's barriers
  if (i >= live) {   // padded agent: its map passes through un-fused
#pragma unroll
    for (int u = 0; u < KSW; ++u) {
      const int ks = ks_first + u;
      store_piece(ks, *reinterpret_cast<const f32x4*>(xrow + 16 * ks),
                  *reinterpret_cast<const f32x4*>(xrow + 16 * ks + 4));
    }
    if constexpr (!kWL) {
      note_range(amax, nan_seen);
      return;
    }
  }

  // neighbour list, in the reference's order: the ego, then j ascending
  int n = 1;
  jl_s[tw][0] = i;   // NW > 1: every wave writes the same values
  for (int j = 0; j < live; ++j)
    if (j != i && (!a.only_v2i || i == 0 || j == 0)) jl_s[tw][n++] = j;
  if constexpr (NW > 1 || kWL) __syncthreads();   // list, layer 2-4 weights and affines (kWL: and W1_ego) visible to every wave
  auto row_of = [&](int k) { return k == 0 ? xrow_r : yrow_of(jl_s[tw][k]); };

  auto frag_from = [&](const f32x4 v0, const f32x4 v1, half8& fh, half8& fl) {
    u32x2 h0, l0, h1, l1;
    split4(v0, h0, l0);
    split4(v1, h1, l1);
    fh = __builtin_bit_cast(half8, u32x4{h0[0], h0[1], h1[0], h1[1]});
    fl = __builtin_bit_cast(half8, u32x4{l0[0], l0[1], l1[0], l1[1]});
  };

  // ---- layer 1 for NG rows at once: acc[g][nt] += W[mat][nt] . row_g.  The k loop runs two k-steps
  // per trip (never unrolled as a whole: hipcc would hoist every fragment load and spill) with the
  // next step's weight fragments and row pieces in flight under the current step's MFMAs.
  auto layer1 = [&](auto ng_c, const Row (&rows)[G], int cnt, int mat, f32x16 (&acc)[G][4]) {
    constexpr int NG = decltype(ng_c)::value;
    const unsigned char* wbase = a.w1 + (size_t)mat * 4 * KS * 2 * 2 * 32 * 16 + (size_t)(lh * 32 + li) * 16;
    // fragment (nt, ks, part) at wbase + (((nt * KS + ks) * 2 + part) * 64) * 16
    half8 wh[kWRing][4], wl[kWRing][4];   // weight fragments: ring of kWRing k-steps, loaded kWAhead k-steps before their MFMAs
    // row pieces: ring of RING k-steps (the maps come from HBM / the far L2); DN_FUSE_RING deepens it for the weight-in-LDS form (A/B)
    constexpr int RING = (kWL && KS % DN_FUSE_RING == 0) ? DN_FUSE_RING : 4;
    f32x4 r0[RING][NG], r1[RING][NG];
    auto wload = [&](int ks, int s) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if constexpr (kWL) {      // the matrix in use is LDS-resident (stage_w1): the same fragment, at LDS latency
          wh[s][nt] = *reinterpret_cast<const half8*>(w1_s + (((nt * KS + ks) * 2 + 0) * 64 + lh * 32 + li) * 16);
          wl[s][nt] = *reinterpret_cast<const half8*>(w1_s + (((nt * KS + ks) * 2 + 1) * 64 + lh * 32 + li) * 16);
        } else {
          wh[s][nt] = *reinterpret_cast<const half8*>(wbase + (size_t)(((nt * KS + ks) * 2 + 0) * 64) * 16);
          wl[s][nt] = *reinterpret_cast<const half8*>(wbase + (size_t)(((nt * KS + ks) * 2 + 1) * 64) * 16);
        }
      }
    };
    auto rload = [&](int ks, int s) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
        if (g < cnt) {
          const int kk = (kAbl & 2) ? 0 : ks;
          r0[s][g] = *reinterpret_cast<const f32x4*>(rows[g].p + kk * rows[g].kss);
          r1[s][g] = *reinterpret_cast<const f32x4*>(rows[g].p + kk * rows[g].kss + rows[g].r1);
        }
    };
    auto mma = [&](int sw, int sr) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
        if (g < cnt) {
          half8 fh, fl;
          if constexpr ((kAbl & 8) != 0) {
            fh = __builtin_bit_cast(half8, r0[sr][g]);
            fl = __builtin_bit_cast(half8, r1[sr][g]);
          } else {
            frag_from(r0[sr][g], r1[sr][g], fh, fl);
          }
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[g][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[sw][nt], fh, acc[g][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[g][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[sw][nt], fl, acc[g][nt], 0, 0, 0);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[g][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[sw][nt], fh, acc[g][nt], 0, 0, 0);
        }
    };
    static_assert(KS % RING == 0, "k loop runs RING k-steps per trip");
#pragma unroll
    for (int q = 0; q < RING - 1; ++q) rload(q, q);
#pragma unroll
    for (int q = 0; q < kWAhead; ++q) wload(q, q);
#pragma unroll 1
    for (int ks0 = 0; ks0 < KS; ks0 += RING) {
#pragma unroll
      for (int u = 0; u < RING; ++u) {
        const int ks = ks0 + u;
        if (ks + RING - 1 < KS) rload(ks + RING - 1, (u + RING - 1) % RING);
        if (ks + kWAhead < KS) wload(ks + kWAhead, (u + kWAhead) & (kWRing - 1));
        __builtin_amdgcn_sched_barrier(0);
        mma(u & (kWRing - 1), u);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- layers 2-4 on acc (= E + F_k) -> exp(s)
  auto tail = [&](f32x16 (&acc)[4]) -> float {
    if constexpr ((kAbl & 4) != 0) return acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      u32x2 hi[4], lo[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int u = nt * 32 + 8 * g + 4 * lh;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(s1l + u), sh = *reinterpret_cast<const f32x4*>(t1l + u);
        float unused = 0.f;
        split4(affine4(quad_of(acc[nt], g), sc, sh), hi[g], lo[g], unused, 0.f);   // ReLU = the split's lower clamp
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int ks = nt * 2 + m;
        const half8 xh = __builtin_bit_cast(half8, gather_octet(hi[2 * m], hi[2 * m + 1]));
        const half8 xl = __builtin_bit_cast(half8, gather_octet(lo[2 * m], lo[2 * m + 1]));
        const half8 wh = frag_of(w2l, ((ks * 2 + 0) * 2 + lh) * 32 + li), wl = frag_of(w2l, ((ks * 2 + 1) * 2 + lh) * 32 + li);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc2, 0, 0, 0);
      }
    }
    f32x16 acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
    {
      u32x2 hi[4], lo[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int u = 8 * g + 4 * lh;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(s2l + u), sh = *reinterpret_cast<const f32x4*>(t2l + u);
        float unused = 0.f;
        split4(affine4(quad_of(acc2, g), sc, sh), hi[g], lo[g], unused, 0.f);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const half8 xh = __builtin_bit_cast(half8, gather_octet(hi[2 * m], hi[2 * m + 1]));
        const half8 xl = __builtin_bit_cast(half8, gather_octet(lo[2 * m], lo[2 * m + 1]));
        const half8 wh = frag_of(w3l, ((m * 2 + 0) * 2 + lh) * 32 + li), wl = frag_of(w3l, ((m * 2 + 1) * 2 + lh) * 32 + li);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc3, 0, 0, 0);
      }
    }
    // units 0..7 of layer 3: register quad 0 of lane (j, h) = units 4h..4h+3
    const f32x4 sc = *reinterpret_cast<const f32x4*>(s3l + 4 * lh), sh = *reinterpret_cast<const f32x4*>(t3l + 4 * lh);
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(w4l + 4 * lh);
    float part = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) part += fmaxf(acc3[e] * sc[e] + sh[e], 0.f) * w4[e];
    // the other four units sit in lane (j, 1 - h); the sum is taken in unit order 0..7 on both lanes
    const float other = __shfl_xor(part, 32, 64);
    const float s = fmaxf((lh ? other + part : part + other) + b4v, 0.f);
    return expf(s);
  };

  // ---- pass 1a: E = W1_ego . x_ego, parked in LDS between the groups ([reg][lane]); kWL: kept in registers
  f32x16 Ereg[kWL ? 4 : 1];
  mark(0);
  if constexpr (kWL) {
    if (active) {
      f32x16 acc[G][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][nt][r] = 0.f;
      Row rows[G];
#pragma unroll
      for (int g = 0; g < G; ++g) rows[g] = xrow_r;
      layer1(std::integral_constant<int, 1>{}, rows, 1, 0, acc);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) Ereg[nt] = acc[0][nt];
#if DN_FUSE_PHASES
      { float probe = Ereg[0][0]; asm volatile("v_mov_b32 %0, %0" : "+v"(probe)); }
#endif
    }
    mark(1);
    __syncthreads();      // every wave is done with W1_ego
    stage_w1(1);
    __syncthreads();      // W1_nbr in place
    mark(0);
  } else if constexpr (NW == 1) {
    f32x16 acc[G][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][nt][r] = 0.f;
    Row rows[G];
#pragma unroll
    for (int g = 0; g < G; ++g) rows[g] = xrow_r;
    layer1(std::integral_constant<int, 1>{}, rows, 1, 0, acc);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) e_s[nt * 16 + r][lane] = acc[0][nt][r];
  } else {
    // unit tile nt = wave of E: per tile the same MFMA sequence as layer1 (wl.fh, wh.fl, wh.fh per k-step)
    static_assert(NW == 1 || NW == 4, "one layer-1 unit tile per wave");
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const unsigned char* wb = a.w1 + (size_t)(lh * 32 + li) * 16 + (size_t)(wave * KS) * 2 * 64 * 16;
    half8 wh[2], wl[2];
    f32x4 r0[4], r1[4];
    auto wload = [&](int ks, int sl) {
      wh[sl] = *reinterpret_cast<const half8*>(wb + (size_t)((ks * 2 + 0) * 64) * 16);
      wl[sl] = *reinterpret_cast<const half8*>(wb + (size_t)((ks * 2 + 1) * 64) * 16);
    };
    auto rload = [&](int ks, int sl) {
      r0[sl] = *reinterpret_cast<const f32x4*>(xrow + 16 * ks);
      r1[sl] = *reinterpret_cast<const f32x4*>(xrow + 16 * ks + 4);
    };
    rload(0, 0);
    rload(1, 1);
    rload(2, 2);
    wload(0, 0);
#pragma unroll 1
    for (int ks0 = 0; ks0 < KS; ks0 += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ks = ks0 + u;
        if (ks + 3 < KS) rload(ks + 3, (u + 3) & 3);
        if (ks + 1 < KS) wload(ks + 1, (u + 1) & 1);
        half8 fh, fl;
        frag_from(r0[u], r1[u], fh, fl);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[u & 1], fh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[u & 1], fl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[u & 1], fh, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) e_s[wave * 16 + r][lane] = acc[r];
    __syncthreads();
  }
  // ---- pass 1b: scores of the list.  NW = 1: G slots at a time; NW = 4: this wave's slots wave, wave + 4, ...
  if constexpr (kWL) {
    if (!active) {      // nothing left for this wave (no tile, or a padded agent already passed through); no barrier follows
      note_range(amax, nan_seen);
      return;
    }
  }
  if constexpr (NW == 1) {
    for (int g0 = 0; g0 < n; g0 += G) {
      const int cnt = n - g0 < G ? n - g0 : G;
      f32x16 acc[G][4];
      Row rows[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        rows[g] = row_of(g0 + g < n ? g0 + g : 0);
        if (g < cnt) {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            if constexpr (kWL) acc[g][nt] = Ereg[nt];
            else {
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[g][nt][r] = e_s[nt * 16 + r][lane];
            }
          }
        }
      }
      layer1(std::integral_constant<int, G>{}, rows, cnt, 1, acc);
#if DN_FUSE_PHASES
      { float probe = acc[0][0][0]; asm volatile("v_mov_b32 %0, %0" : "+v"(probe)); }
#endif
      mark(2);
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (g < cnt) ek_s[tw][g0 + g][lane] = tail(acc[g]);
      mark(3);
    }
  } else {
    for (int k = wave; k < n; k += NW) {   // one slot per pass: two accumulator sets would spill at 256 registers
      f32x16 acc[G][4];
      Row rows[G];
#pragma unroll
      for (int g = 0; g < G; ++g) rows[g] = row_of(k);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][nt][r] = e_s[nt * 16 + r][lane];
      layer1(std::integral_constant<int, 1>{}, rows, 1, 1, acc);
      ek_s[0][k][lane] = tail(acc[0]);
    }
    __syncthreads();
  }
  float den = 0.f;
  for (int k = 0; k < n; ++k) den += ek_s[tw][k][lane];

  // ---- pass 2: weighted sum in list order; the next slot's row is in flight under the FMAs
  f32x4 f0[KSW], f1[KSW], y0[2][KSW], y1[2][KSW];
  auto yload = [&](int k, int s) {
    const Row row = row_of(k);
    if constexpr ((kAbl & 1) != 0) {
#pragma unroll
      for (int ks = 0; ks < KSW; ++ks) y0[s][ks] = y1[s][ks] = f32x4{1.f, 2.f, 3.f, (float)k};
      return;
    }
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) {
      y0[s][ks] = *reinterpret_cast<const f32x4*>(row.p + (ks_first + ks) * row.kss);
      y1[s][ks] = *reinterpret_cast<const f32x4*>(row.p + (ks_first + ks) * row.kss + row.r1);
    }
  };
  auto yacc = [&](int k, int s) {
    const float w = ek_s[tw][k][lane] / den;
    if (a.weights_out && pvalid && lh == 0 && wave == 0)
      a.weights_out[(((size_t)b * a.ego_count + il) * a.agents + k) * a.hw + p] = w;
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) {
      f0[ks] += y0[s][ks] * w;
      f1[ks] += y1[s][ks] * w;
    }
  };
#pragma unroll
  for (int ks = 0; ks < KSW; ++ks) f0[ks] = f1[ks] = f32x4{0.f, 0.f, 0.f, 0.f};
  yload(0, 0);
  for (int k = 0; k < n; k += 2) {
    if (k + 1 < n) yload(k + 1, 1);
    yacc(k, 0);
    if (k + 1 < n) {
      if (k + 2 < n) yload(k + 2, 0);
      yacc(k + 1, 1);
    }
  }
#pragma unroll
  for (int ks = 0; ks < KSW; ++ks) store_piece(ks_first + ks, f0[ks], f1[ks]);
  note_range(amax, nan_seen);
#if DN_FUSE_PHASES
  mark(4);
  ph[5] = 1;
  ph[6] = __builtin_readcyclecounter() - t_begin;
  if (lane == 0)
    for (int k = 0; k < 7; ++k) atomicAdd(&g_fuse_phase[k], ph[k]);
#endif
}

// weights [rows][cols] * wmul -> A-operand fragments [nt][ks][part][h][32 rows] x 16 B (rows / cols
// past the matrix are zero); `col0` = first column of the source matrix (row stride ld)
__global__ void pack_frags_kernel(const float* __restrict__ w, int ld, int col0, int rows, int cols,
                                  int n_tiles, int ksteps, float wmul, unsigned char* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // (nt, ks, h, row)
  if (idx >= n_tiles * ksteps * 2 * 32) return;
  const int row = idx % 32, h = (idx / 32) % 2, ks = (idx / 64) % ksteps, nt = idx / (64 * ksteps);
  half8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int r = nt * 32 + row, c = ks * 16 + h * 8 + e;
    float v = (r < rows && c < cols) ? w[(size_t)r * ld + col0 + c] * wmul : 0.f;
    v = fminf(fmaxf(v, -65504.f), 65504.f);
    hi[e] = (_Float16)v;
    lo[e] = (_Float16)(v - (float)hi[e]);
  }
  *reinterpret_cast<half8*>(out + (size_t)((((nt * ksteps + ks) * 2 + 0) * 2 + h) * 32 + row) * 16) = hi;
  *reinterpret_cast<half8*>(out + (size_t)((((nt * ksteps + ks) * 2 + 1) * 2 + h) * 32 + row) * 16) = lo;
}

inline size_t w1_bytes(int c) { return (size_t)2 * 4 * (c / 16) * 2 * 2 * 32 * 16; }

}  // namespace

namespace dn { void range_collect_fuse_mlp(unsigned* dst, bool reset, hipStream_t s) { sp_range_collect_here(dst, reset, s); } }

int g_fuse_waves = 0;   // 0 = DN_FUSE_MLP_WAVES, else chosen per launch

// tools / tests: the launch form -- 1 or 4 waves per 32-pixel tile, 2 = the weight-in-LDS form (0 = default)
extern "C" int dn_fuse_mlp_set_waves(int waves) {
  DN_REQUIRE(waves == 0 || waves == 1 || waves == 2 || waves == 4,
             "fuse_mlp: form is 1 or 4 waves per tile, 2 = layer-1 weights in LDS (0 = default), got %d", waves);
  g_fuse_waves = waves;
  return DN_OK;
}

extern "C" int dn_fuse_mlp_supported(int c) { return c == 64 || c == 128 || c == 256; }

extern "C" size_t dn_fuse_mlp_packed_bytes(int c) { return w1_bytes(c) + kW2Bytes + kW3Bytes; }

extern "C" int dn_fuse_mlp_pack(const float* w1, const float* w2, const float* w3, int c, float wmul1,
                                float wmul2, float wmul3, void* packed, void* stream) {
  DN_REQUIRE(w1 && w2 && w3 && packed, "fuse_mlp pack: null pointer");
  DN_REQUIRE(dn_fuse_mlp_supported(c), "fuse_mlp pack: %d channels unsupported (64, 128 or 256)", c);
  hipStream_t s = (hipStream_t)stream;
  unsigned char* out = (unsigned char*)packed;
  const int ks = c / 16;
  // W1 [128][2C] = [W_ego | W_nbr]
  for (int mat = 0; mat < 2; ++mat)
    hipLaunchKernelGGL(pack_frags_kernel, dim3((4 * ks * 64 + 255) / 256), dim3(256), 0, s, w1, 2 * c, mat * c,
                       128, c, 4, ks, wmul1, out + (size_t)mat * (w1_bytes(c) / 2));
  hipLaunchKernelGGL(pack_frags_kernel, dim3((8 * 64 + 255) / 256), dim3(256), 0, s, w2, 128, 0, 32, 128, 1, 8,
                     wmul2, out + w1_bytes(c));
  hipLaunchKernelGGL(pack_frags_kernel, dim3((2 * 64 + 255) / 256), dim3(256), 0, s, w3, 32, 0, 8, 32, 1, 2,
                     wmul3, out + w1_bytes(c) + kW2Bytes);
  return dn::check_launch("pack_frags_kernel");
}

namespace {
// the weight-in-LDS form of <C, WL tiles per workgroup>: a FUNCTION template, so that every instantiation has its own
// per-device "attribute set" flag (a generic lambda taking the kernel pointer instantiates ONCE for all nine kernels -- they
// share the type void (*)(FuseMlpArgs) -- and raised the dynamic-LDS limit of the first variant launched only)
template <int CC, int WL>
int launch_wl(const FuseMlpArgs& a, dim3 grid, int lds, hipStream_t s) {
  auto kern = disco_fuse_mlp_kernel<CC, FUSE_GL, 1, WL>;
  static dn::PerDeviceFlag flag;
  bool& done = flag.here();
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return dn::fail(DN_ERR_LAUNCH, "fuse_mlp: cannot reserve %d B of dynamic LDS", lds);
    done = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * WL), lds, s, a);
  return dn::check_launch("disco_fuse_mlp_kernel (weights in LDS)");
}

int fuse_mlp_impl(const float* feat, const float* warped, int warped_fm, const int32_t* num_agent, const dn_fuse_mlp_params* p,
                  int batch, int agents, int hw, int c, int only_v2i, int ego_first, int ego_count, void* fused_sp,
                  float* fused_nhwc, float* weights_out, void* stream);
}

extern "C" int dn_disco_fuse_mlp(const float* feat, const float* warped, const int32_t* num_agent,
                                 const dn_fuse_mlp_params* p, int batch, int agents, int hw, int c,
                                 int only_v2i, int ego_first, int ego_count, void* fused_sp,
                                 float* fused_nhwc, float* weights_out, void* stream) {
  return fuse_mlp_impl(feat, warped, 0, num_agent, p, batch, agents, hw, c, only_v2i, ego_first, ego_count, fused_sp,
                       fused_nhwc, weights_out, stream);
}

// `warped` in the fragment-major form dn_warp_neighbors_fm writes (hw % 32 == 0): identical results, the neighbour
// rows arrive as contiguous 1 KB runs per load instruction instead of 32 half cache lines.
extern "C" int dn_disco_fuse_mlp_fm(const float* feat, const float* warped_fm, const int32_t* num_agent,
                                    const dn_fuse_mlp_params* p, int batch, int agents, int hw, int c,
                                    int only_v2i, int ego_first, int ego_count, void* fused_sp,
                                    float* fused_nhwc, float* weights_out, void* stream) {
  DN_REQUIRE(hw % 32 == 0, "fuse_mlp (fragment-major): hw %d must be a multiple of 32", hw);
  return fuse_mlp_impl(feat, warped_fm, 1, num_agent, p, batch, agents, hw, c, only_v2i, ego_first, ego_count, fused_sp,
                       fused_nhwc, weights_out, stream);
}

namespace {
int fuse_mlp_impl(const float* feat, const float* warped, int warped_fm, const int32_t* num_agent, const dn_fuse_mlp_params* p,
                  int batch, int agents, int hw, int c, int only_v2i, int ego_first, int ego_count, void* fused_sp,
                  float* fused_nhwc, float* weights_out, void* stream) {
  DN_REQUIRE(feat && num_agent && p && (fused_sp || fused_nhwc), "fuse_mlp: null pointer");
  DN_REQUIRE(agents < 2 || warped, "fuse_mlp: neighbours present but warped is null");
  DN_REQUIRE(batch > 0 && agents > 0 && hw > 0, "fuse_mlp: empty problem");
  DN_REQUIRE(dn_fuse_mlp_supported(c), "fuse_mlp: %d channels unsupported (64, 128 or 256)", c);
  DN_REQUIRE(ego_first >= 0 && ego_count > 0 && ego_first + ego_count <= agents,
             "fuse_mlp: ego range [%d, %d) outside 0..%d", ego_first, ego_first + ego_count, agents);
  DN_REQUIRE(agents <= MAX_AGENTS, "fuse_mlp: at most %d agents supported (got %d)", MAX_AGENTS, agents);
  DN_REQUIRE(p->packed && p->s1 && p->t1 && p->s2 && p->t2 && p->s3 && p->t3 && p->w4 && p->b4,
             "fuse_mlp: null MLP parameter");
  auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  DN_REQUIRE(aligned16(feat) && aligned16(warped) && aligned16(p->packed) && aligned16(fused_sp) &&
                 aligned16(fused_nhwc) && aligned16(p->s1) && aligned16(p->t1) && aligned16(p->s2) &&
                 aligned16(p->t2) && aligned16(p->s3) && aligned16(p->t3) && aligned16(p->w4),
             "fuse_mlp: buffers must be 16-byte aligned");
  FuseMlpArgs a;
  a.feat = feat; a.warped = warped; a.num_agent = num_agent;
  a.w1 = (const unsigned char*)p->packed;
  a.w2 = a.w1 + w1_bytes(c);
  a.w3 = a.w2 + kW2Bytes;
  a.s1 = p->s1; a.t1 = p->t1; a.s2 = p->s2; a.t2 = p->t2; a.s3 = p->s3; a.t3 = p->t3; a.w4 = p->w4; a.b4 = p->b4;
  a.fused_sp = (unsigned char*)fused_sp; a.fused_nhwc = fused_nhwc; a.weights_out = weights_out;
  a.batch = batch; a.agents = agents; a.hw = hw; a.only_v2i = only_v2i;
  a.ego_first = ego_first; a.ego_count = ego_count;
  a.tiles = (hw + 31) / 32;
  a.total_tiles = batch * ego_count * a.tiles;
  a.warped_fm = warped_fm;
  dim3 grid(a.total_tiles);   // one workgroup per 32 pixels of one (sample, ego)
  hipStream_t s = (hipStream_t)stream;
  // Three forms, bit-identical results (tests/test_gpu_fusion.py):
  //   4 -- four waves per tile, when the launch leaves SIMDs idle (fewer tiles than 2 per CU: 128 tiles for one rank's
  //        share of the agent-sharded step, 87 -> 45 us);
  //   2 -- layer-1 weights staged in LDS once per workgroup of WL tiles (round 5), from 2 tiles per CU up (the BASELINE
  //        size: 640 tiles -> 214 workgroups of three one-wave chains);
  //   1 -- one wave per tile streaming its weight fragments from L2 (rounds 2-4's form at the BASELINE size; kept for A/B).
  // DN_FUSE_MLP_WAVES / dn_fuse_mlp_set_waves force a form (tools, tests).
  static const int waves_env = [] { const char* e = getenv("DN_FUSE_MLP_WAVES"); return e ? atoi(e) : 0; }();
  const int forced = g_fuse_waves > 0 ? g_fuse_waves : waves_env;
  const int waves = forced > 0 ? forced : (grid.x < 2u * 256u ? 4 : 2);
  if (waves == 2) {
    // workgroups of WL tiles so that the launch is one resident generation (one workgroup per CU: 128 KB of LDS at C = 256)
    const int wl = a.total_tiles <= 2 * 256 ? 2 : a.total_tiles <= 3 * 256 ? 3 : 4;
    const dim3 g2((a.total_tiles + wl - 1) / wl);
    const int lds = 4 * (c / 16) * 2 * 2 * 32 * 16;
    int rc = DN_OK;
#define DN_FUSE_WL(CC)                                                                        \
    rc = wl == 2 ? launch_wl<CC, 2>(a, g2, lds, s) : wl == 3 ? launch_wl<CC, 3>(a, g2, lds, s) : launch_wl<CC, 4>(a, g2, lds, s)
    if (c == 256) DN_FUSE_WL(256);
    else if (c == 128) DN_FUSE_WL(128);
    else DN_FUSE_WL(64);
#undef DN_FUSE_WL
    if (rc) return rc;
  } else if (waves == 1) {
    if (c == 256) hipLaunchKernelGGL((disco_fuse_mlp_kernel<256, FUSE_G, 1>), grid, dim3(64), 0, s, a);
    else if (c == 128) hipLaunchKernelGGL((disco_fuse_mlp_kernel<128, FUSE_G, 1>), grid, dim3(64), 0, s, a);
    else hipLaunchKernelGGL((disco_fuse_mlp_kernel<64, FUSE_G, 1>), grid, dim3(64), 0, s, a);
  } else {
    if (c == 256) hipLaunchKernelGGL((disco_fuse_mlp_kernel<256, 1, 4>), grid, dim3(256), 0, s, a);
    else if (c == 128) hipLaunchKernelGGL((disco_fuse_mlp_kernel<128, 1, 4>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((disco_fuse_mlp_kernel<64, 1, 4>), grid, dim3(256), 0, s, a);
  }
  return dn::check_launch("disco_fuse_mlp_kernel");
}
}  // namespace
