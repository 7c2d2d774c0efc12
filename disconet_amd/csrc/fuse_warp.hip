// K4 + K5 + K6 in ONE launch: the pose warp of the neighbour maps, the pairwise attention MLP, the softmax over
// the agents and the weighted sum -- the whole DiscoGraph fusion of a (sample, ego) pixel tile without a single
// intermediate tensor: the warped neighbour maps (84 MB per step at the BASELINE shape, written by
// dn_warp_neighbors and read twice by dn_disco_fuse_mlp) never exist.
//
// Replaces feature_transformation (upstream:coperception/models/det/base/*), PixelWeightedFusionSoftmax.forward
// and the fusion loop body of upstream:coperception/models/det/DiscoNet.py :: DiscoNet.forward
// (Appx A.4, A.5).  Arithmetic: the two bilinear passes as warp.hip's tile-shared
// form (same taps, same order), the MLP as fuse_mlp.hip (split-f16 x3 MFMA, fp32 accumulate, same packed
// weights), exp without max-shift, list-order sums.
//
// Work decomposition.  A workgroup (4 waves) owns an 8 x 4 pixel tile of one (sample, ego):
//   * the MFMA pixel tile is those 32 pixels (lane j = pixel, lane half h = k octet); the four waves split the
//     128 units of layer 1: wave w owns units 32 w .. 32 w + 31 (one 32 x 32 accumulator per list slot), so a
//     slot's layer-1 work is spread over the four SIMDs of the CU -- 4 x the waves of fuse_mlp.hip, which is
//     latency-bound at one wave per 32 pixels;
//   * the K loop over the C channels runs in 64-channel phases.  Per (neighbour, phase) the workgroup
//       rotate:  computes the rotated map R(q) of the (8 + 2) x (4 + 2) block of pixels q that the tile's
//                translation taps can reach, 64 channels, into LDS (4 source taps each, once per block);
//       blend:   wave w blends k-step w of the phase for its 32 pixels from LDS (the translation pass), splits
//                the 8 channels of each lane into hi / lo f16 and publishes the two 16-byte B fragments in LDS;
//       mfma:    every wave multiplies the phase's four fragment pairs by ITS 32 units' weights (12 MFMAs);
//     pipelined over stages s = (neighbour, phase): between two barriers every wave blends stage s + 1, multiplies
//     stage s, writes the rotated block of stage s + 2 (whose source-tap loads were issued a stage earlier and
//     waited in registers: they come from far L2 / MALL) and issues the loads of stage s + 3; rotated blocks and
//     fragments double-buffered -- one barrier per stage;
//   * a slot's tail (layers 2-4): every wave turns its 32 layer-1 units into two k-steps of layer 2 and leaves a
//     PARTIAL 32 x 32 accumulator in LDS; after the next barrier all waves add the four partials (fixed order)
//     and finish layers 3, 4 and exp redundantly -- no second hand-off;
//   * pass 2 (the weighted sum) re-derives the warped values with the same rotate / blend pipeline (the source
//     maps are L2-resident) and accumulates w_k * y_k in registers: lane (pixel, octet) of wave w keeps the 8
//     channels of k-step w of every phase; written as split-planar pieces and / or fp32 NHWC rows.
#include "dn_internal.h"
#include "sp_device.h"
#include "warp_device.h"

namespace {

constexpr int FW_MAX_AGENTS = 8;
constexpr int FT_W = 8, FT_H = 4;                               // pixel tile
constexpr int FQ_W = FT_W + 2, FQ_H = FT_H + 2, FQ = FQ_W * FQ_H;   // rotated block
constexpr int ROT_F4 = 17;                                      // float4 per rotated pixel: 64 channels + 1 pad
constexpr size_t kW2Bytes = 8 * 2 * 2 * 32 * 16, kW3Bytes = 2 * 2 * 2 * 32 * 16;

struct FuseWarpArgs {
  const float* feat;
  const float* trans;
  const int32_t* num_agent;
  const unsigned char* w1;   // [mat 2 (ego, nbr)][nt 4][ks C/16][part 2][h 2][row 32] x 16 B   (fuse_mlp.hip's image)
  const unsigned char* w2;   // [ks 8][part 2][h 2][row 32] x 16 B, followed by w3 [ks 2][part 2][h 2][row 32]
  const float *s1, *t1, *s2, *t2, *s3, *t3, *w4, *b4;
  unsigned char* fused_sp;
  float* fused_nhwc;
  float* weights_out;
  int batch, agents, h, w, only_v2i, ego_first, ego_count, tiles_x, tiles;
};

template <int C>
__global__ void __launch_bounds__(256, 2) disco_fuse_warp_kernel(const FuseWarpArgs a) {
  constexpr int KS = C / 16, NPH = C / 64;
  __shared__ f32x4 rot_s[2][FQ][ROT_F4];                                   // rotated blocks (also: the tail's partials)
  __shared__ __attribute__((aligned(16))) unsigned char frag_s[2][4][2][64][16];   // [stage][k-step][hi, lo][lane]
  __shared__ __attribute__((aligned(16))) float aff_s[2 * 128 + 2 * 32 + 3 * 8];
  __shared__ float red_s[4][16][64];                                       // layer-2 partials: [wave][reg][lane]
  __shared__ float ek_s[FW_MAX_AGENTS][32];                                // exp(s_k) per list slot and pixel
  __shared__ int jl_s[FW_MAX_AGENTS];
  __shared__ __attribute__((aligned(16))) float pose_s[FW_MAX_AGENTS][8];                               // r00 r01 r10 r11 xt yt, (qx0, qy0) as int bits

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int hw = a.h * a.w;
  const int tile = blockIdx.x % a.tiles, il = (blockIdx.x / a.tiles) % a.ego_count, b = blockIdx.x / (a.tiles * a.ego_count);
  const int i = a.ego_first + il;
  const int tile_x0 = (tile % a.tiles_x) * FT_W, tile_y0 = (tile / a.tiles_x) * FT_H;
  const int px = tile_x0 + (li & 7), py = tile_y0 + (li >> 3);
  const bool pvalid = px < a.w && py < a.h;
  const int p = pvalid ? py * a.w + px : 0;                                // clamped for loads
  int live = a.num_agent[b];
  live = live < 0 ? 0 : (live < a.agents ? live : a.agents);
  const size_t oimg = (size_t)il * a.batch + b;
  const float* xrow = a.feat + (((size_t)i * a.batch + b) * hw + p) * C + 8 * lh;   // this lane's ego row, octet lh of a k-step

  float amax = 0.f;
  auto store_piece = [&](int ks, const f32x4 v0, const f32x4 v1) {        // 8 channels of k-step ks of this lane's pixel
    if (!pvalid) return;
    if (a.fused_sp) {
      u32x2 h0, l0, h1, l1;
      split4(v0, h0, l0, amax);
      split4(v1, h1, l1, amax);
      const size_t plane = (size_t)hw * 16;
      unsigned char* o = a.fused_sp + ((oimg * KS + ks) * 4 + lh) * plane + (size_t)p * 16;
      *reinterpret_cast<u32x4*>(o) = u32x4{h0[0], h0[1], h1[0], h1[1]};
      *reinterpret_cast<u32x4*>(o + 2 * plane) = u32x4{l0[0], l0[1], l1[0], l1[1]};
    }
    if (a.fused_nhwc) {
      float* o = a.fused_nhwc + (oimg * hw + p) * C + 16 * ks + 8 * lh;
      *reinterpret_cast<f32x4*>(o) = v0;
      *reinterpret_cast<f32x4*>(o + 4) = v1;
    }
  };

  if (i >= live) {   // padded agent: its map passes through un-fused (wave w copies k-steps w, w + 4, ...)
    for (int ks = wave; ks < KS; ks += 4)
      store_piece(ks, *reinterpret_cast<const f32x4*>(xrow + 16 * ks), *reinterpret_cast<const f32x4*>(xrow + 16 * ks + 4));
    note_range(amax);
    return;
  }

  // ---- layers 2-4: affines in LDS; the weight fragments (4 per wave and tail half) come straight from L2 --
  // 20 KB of LDS for them would cost the second workgroup per CU
  if (tid < 128) { aff_s[tid] = a.s1[tid]; aff_s[128 + tid] = a.t1[tid]; }
  if (tid < 32) { aff_s[256 + tid] = a.s2[tid]; aff_s[288 + tid] = a.t2[tid]; }
  if (tid < 8) { aff_s[320 + tid] = a.s3[tid]; aff_s[328 + tid] = a.t3[tid]; aff_s[336 + tid] = a.w4[tid]; }
  const float b4v = a.b4[0];
  const unsigned char* w2l = a.w2;
  const unsigned char* w3l = a.w2 + kW2Bytes;
  const float *s1l = aff_s, *t1l = aff_s + 128, *s2l = aff_s + 256, *t2l = aff_s + 288, *s3l = aff_s + 320,
              *t3l = aff_s + 328, *w4l = aff_s + 336;

  // neighbour list in the reference's order: the ego, then j ascending (every thread builds the same list)
  int n = 1;
  if (tid == 0) jl_s[0] = i;
  for (int j = 0; j < live; ++j)
    if (j != i && (!a.only_v2i || i == 0 || j == 0)) {
      if (tid == 0) jl_s[n] = j;
      ++n;
    }
  __syncthreads();
  const int n_stages = (n - 1) * NPH;

  auto frag_from = [&](const f32x4 v0, const f32x4 v1, half8& fh, half8& fl) {
    u32x2 h0, l0, h1, l1;
    split4(v0, h0, l0);
    split4(v1, h1, l1);
    fh = __builtin_bit_cast(half8, u32x4{h0[0], h0[1], h1[0], h1[1]});
    fl = __builtin_bit_cast(half8, u32x4{l0[0], l0[1], l1[0], l1[1]});
  };
  // layer-1 weight fragment (mat, this wave's unit tile, ks, part) of this lane
  const unsigned char* w1base = a.w1 + (size_t)(lh * 32 + li) * 16;
  auto w1frag = [&](int mat, int ks, int part) {
    return *reinterpret_cast<const half8*>(w1base + ((size_t)mat * 4 * KS * 2 + ((size_t)wave * KS + ks) * 2 + part) * 64 * 16);
  };

  // ---- pose of stage s = (slot k = 1 + s / NPH, phase s % NPH) and the rotated block of its tile
  struct Pose {
    float r00, r01, r10, r11, xt, yt;
    int qx0, qy0;
    const float* src;
  };
  // computed once per workgroup (thread k - 1 for list slot k) into pose_s: a stage reads it from LDS instead of
  // chasing jl_s -> trans -> 6 dependent global loads three times per stage
  auto pose_build = [&](int k) {
    const int j = jl_s[k];
    const float* m = a.trans + (((size_t)b * a.agents + i) * a.agents + j) * 16;
    const float xt = (4.f * m[3]) / 128.f, yt = -(4.f * m[7]) / 128.f;
    // north-west q of the tile: the smallest x0(p) - (p - tile origin) over the tile's columns / rows
    int qx = 1 << 30, qy = 1 << 30;
#pragma unroll
    for (int kk = 0; kk < FT_W; ++kk) {
      const Bilinear t = bilinear_taps((2.f * (tile_x0 + kk) + 1.f) / a.w - 1.f + xt,
                                       (2.f * (tile_y0 + (kk < FT_H ? kk : 0)) + 1.f) / a.h - 1.f + yt, a.w, a.h);
      qx = min(qx, t.x0 - kk);
      if (kk < FT_H) qy = min(qy, t.y0 - kk);
    }
    pose_s[k][0] = m[0]; pose_s[k][1] = m[1]; pose_s[k][2] = m[4]; pose_s[k][3] = m[5];
    pose_s[k][4] = xt; pose_s[k][5] = yt;
    pose_s[k][6] = __int_as_float(qx); pose_s[k][7] = __int_as_float(qy);
  };
  if (tid >= 1 && tid < n) pose_build(tid);
  __syncthreads();
  auto pose_of = [&](int k) {
    Pose ps;
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(&pose_s[k][0]), p1 = *reinterpret_cast<const f32x4*>(&pose_s[k][4]);
    ps.r00 = p0[0]; ps.r01 = p0[1]; ps.r10 = p0[2]; ps.r11 = p0[3];
    ps.xt = p1[0]; ps.yt = p1[1];
    ps.qx0 = __float_as_int(p1[2]); ps.qy0 = __float_as_int(p1[3]);
    ps.src = a.feat + ((size_t)jl_s[k] * a.batch + b) * hw * C;
    return ps;
  };
  // rotate: R(q) for the FQ pixels of the block, channels [64 ph, 64 ph + 64), into rot_s[buf] -- in two halves so
  // that the 16 tap loads of a thread (far L2 / MALL: ~1 us) are in flight during a whole stage of the pipeline:
  //   rotate_issue(s, v):        the clamped tap loads of this thread's (up to) 4 block pixels into registers
  //   rotate_commit(s, buf, v):  sample_src's blend (same masked weights, same order) and the LDS write
  constexpr int RIT = (FQ * 16 + 255) / 256;
  struct RotTaps {
    f32x4 v[RIT][4];
  };
  auto rot_taps_of = [&](const Pose& ps, int idx) {
    const int q = min(idx >> 4, FQ - 1);
    const int qx = ps.qx0 + q % FQ_W, qy = ps.qy0 + q / FQ_W;
    const float qbx = (2.f * qx + 1.f) / a.w - 1.f;
    const float qby = (2.f * qy + 1.f) / a.h - 1.f;
    return bilinear_taps(ps.r00 * qbx + ps.r01 * qby, ps.r10 * qbx + ps.r11 * qby, a.w, a.h);
  };
  auto rotate_issue = [&](int s, RotTaps& rt) {
    const int k = 1 + s / NPH, ph = s % NPH;
    const Pose ps = pose_of(k);
    const SrcImage src = make_src_image(ps.src, (size_t)hw * C * 4);
    const unsigned lane_off = 16u * (ph * 16 + (tid & 15));
#pragma unroll
    for (int it = 0; it < RIT; ++it) {
      const Bilinear b = rot_taps_of(ps, tid + 256 * it);
      const int x0 = min(max(b.x0, 0), a.w - 1), x1 = min(max(b.x0 + 1, 0), a.w - 1);
      const int y0 = min(max(b.y0, 0), a.h - 1), y1 = min(max(b.y0 + 1, 0), a.h - 1);
      rt.v[it][0] = ldb4(src, (unsigned)((y0 * a.w + x0) * C) * 4u + lane_off);
      rt.v[it][1] = ldb4(src, (unsigned)((y0 * a.w + x1) * C) * 4u + lane_off);
      rt.v[it][2] = ldb4(src, (unsigned)((y1 * a.w + x0) * C) * 4u + lane_off);
      rt.v[it][3] = ldb4(src, (unsigned)((y1 * a.w + x1) * C) * 4u + lane_off);
    }
  };
  auto rotate_commit = [&](int s, int buf, const RotTaps& rt) {
    const Pose ps = pose_of(1 + s / NPH);
#pragma unroll
    for (int it = 0; it < RIT; ++it) {
      const int idx = tid + 256 * it;
      const Bilinear b = rot_taps_of(ps, idx);
      const bool x0ok = b.x0 >= 0 && b.x0 < a.w, x1ok = b.x0 + 1 >= 0 && b.x0 + 1 < a.w;
      const bool y0ok = b.y0 >= 0 && b.y0 < a.h, y1ok = b.y0 + 1 >= 0 && b.y0 + 1 < a.h;
      f32x4 acc = rt.v[it][0] * ((y0ok && x0ok) ? b.w_nw : 0.f);     // order of torch's CPU kernel: nw, ne, sw, se
      acc += rt.v[it][1] * ((y0ok && x1ok) ? b.w_ne : 0.f);
      acc += rt.v[it][2] * ((y1ok && x0ok) ? b.w_sw : 0.f);
      acc += rt.v[it][3] * ((y1ok && x1ok) ? b.w_se : 0.f);
      if (idx < FQ * 16) rot_s[buf][idx >> 4][tid & 15] = acc;
    }
  };
  // blend: the translation pass for this lane's pixel, channels of k-step `wave` of the phase, octet lh
  auto blend = [&](int s, int buf, f32x4& v0, f32x4& v1) {
    const int k = 1 + s / NPH;
    const Pose ps = pose_of(k);
    const float bx = (2.f * px + 1.f) / a.w - 1.f;
    const float by = (2.f * py + 1.f) / a.h - 1.f;
    const Bilinear t2 = bilinear_taps(bx + ps.xt, by + ps.yt, a.w, a.h);
    const float qw[4] = {t2.w_nw, t2.w_ne, t2.w_sw, t2.w_se};
    const int l0 = wave * 4 + lh * 2;
    v0 = f32x4{0.f, 0.f, 0.f, 0.f};
    v1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int qx = t2.x0 + (kk & 1), qy = t2.y0 + (kk >> 1);
      const bool qok = qx >= 0 && qx < a.w && qy >= 0 && qy < a.h;
      const int lx = min(max(qx - ps.qx0, 0), FQ_W - 1), ly = min(max(qy - ps.qy0, 0), FQ_H - 1);
      const float wk = qok ? qw[kk] : 0.f;
      v0 += rot_s[buf][ly * FQ_W + lx][l0] * wk;
      v1 += rot_s[buf][ly * FQ_W + lx][l0 + 1] * wk;
    }
  };
  auto publish = [&](int buf, const f32x4 v0, const f32x4 v1) {
    half8 fh, fl;
    frag_from(v0, v1, fh, fl);
    *reinterpret_cast<half8*>(&frag_s[buf][wave][0][lane][0]) = fh;
    *reinterpret_cast<half8*>(&frag_s[buf][wave][1][lane][0]) = fl;
  };

  // ---- tail of a slot, first half: this wave's 32 layer-1 units -> two k-steps of layer 2 -> partial in LDS
  auto tail_partial = [&](const f32x16& acc) {
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    u32x2 hi[4], lo[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int u = wave * 32 + 8 * g + 4 * lh;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(s1l + u), sh = *reinterpret_cast<const f32x4*>(t1l + u);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[4 * g + e] * sc[e] + sh[e], 0.f);
      split4(v, hi[g], lo[g]);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int ks = wave * 2 + m;
      const half8 xh = __builtin_bit_cast(half8, gather_octet(hi[2 * m], hi[2 * m + 1]));
      const half8 xl = __builtin_bit_cast(half8, gather_octet(lo[2 * m], lo[2 * m + 1]));
      const half8 wh = *reinterpret_cast<const half8*>(w2l + (size_t)(((ks * 2 + 0) * 2 + lh) * 32 + li) * 16);
      const half8 wl = *reinterpret_cast<const half8*>(w2l + (size_t)(((ks * 2 + 1) * 2 + lh) * 32 + li) * 16);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc2, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red_s[wave][r][lane] = acc2[r];
  };
  // second half (after a barrier): add the four partials in wave order, layers 3 and 4, exp -> ek_s[k]
  auto tail_finish = [&](int k) {
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = ((red_s[0][r][lane] + red_s[1][r][lane]) + red_s[2][r][lane]) + red_s[3][r][lane];
    f32x16 acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
    u32x2 hi[4], lo[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int u = 8 * g + 4 * lh;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(s2l + u), sh = *reinterpret_cast<const f32x4*>(t2l + u);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc2[4 * g + e] * sc[e] + sh[e], 0.f);
      split4(v, hi[g], lo[g]);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const half8 xh = __builtin_bit_cast(half8, gather_octet(hi[2 * m], hi[2 * m + 1]));
      const half8 xl = __builtin_bit_cast(half8, gather_octet(lo[2 * m], lo[2 * m + 1]));
      const half8 wh = *reinterpret_cast<const half8*>(w3l + (size_t)(((m * 2 + 0) * 2 + lh) * 32 + li) * 16);
      const half8 wl = *reinterpret_cast<const half8*>(w3l + (size_t)(((m * 2 + 1) * 2 + lh) * 32 + li) * 16);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc3, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc3, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc3, 0, 0, 0);
    }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(s3l + 4 * lh), sh = *reinterpret_cast<const f32x4*>(t3l + 4 * lh);
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(w4l + 4 * lh);
    float part = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) part += fmaxf(acc3[e] * sc[e] + sh[e], 0.f) * w4[e];
    const float other = __shfl_xor(part, 32, 64);
    const float sk = fmaxf((lh ? other + part : part + other) + b4v, 0.f);
    if (wave == 0 && lh == 0) ek_s[k][li] = expf(sk);     // every wave computed the same value: one writes
  };

  // ---- pass 1a: E = W1_ego . x_ego and F_0 = W1_nbr . x_ego (the ego is list slot 0) -- the ego rows are read
  // where they lie (an NHWC row piece of 8 floats is a B fragment after one split), every wave its own copy
  f32x16 accE, acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) accE[r] = acc[r] = 0.f;
#pragma unroll 1
  for (int ph = 0; ph < NPH; ++ph) {     // a phase's 8 row pieces + 16 weight fragments in flight, then its 24 MFMAs
    f32x4 r0[4], r1[4];
    half8 weh[4], wel[4], wnh[4], wnl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ks = ph * 4 + u;
      r0[u] = *reinterpret_cast<const f32x4*>(xrow + 16 * ks);
      r1[u] = *reinterpret_cast<const f32x4*>(xrow + 16 * ks + 4);
      weh[u] = w1frag(0, ks, 0); wel[u] = w1frag(0, ks, 1);
      wnh[u] = w1frag(1, ks, 0); wnl[u] = w1frag(1, ks, 1);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      half8 fh, fl;
      frag_from(r0[u], r1[u], fh, fl);
      accE = __builtin_amdgcn_mfma_f32_32x32x16_f16(wel[u], fh, accE, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wnl[u], fh, acc, 0, 0, 0);
      accE = __builtin_amdgcn_mfma_f32_32x32x16_f16(weh[u], fl, accE, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wnh[u], fl, acc, 0, 0, 0);
      accE = __builtin_amdgcn_mfma_f32_32x32x16_f16(weh[u], fh, accE, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wnh[u], fh, acc, 0, 0, 0);
    }
  }
  {
    f32x16 sum;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum[r] = accE[r] + acc[r];
    tail_partial(sum);
  }
  int pending_tail = 0;        // list slot whose partials sit in red_s

  // ---- pass 1b: the neighbours.  Stage pipeline (one barrier per stage): after barrier(s) every wave blends
  // stage s + 1 from its rotated block into fragments, multiplies stage s, commits the rotated block of stage
  // s + 2 (its tap loads were issued a stage ago) and issues the tap loads of stage s + 3.
  RotTaps rt;
  if (n_stages > 0) {
    rotate_issue(0, rt);
    rotate_commit(0, 0, rt);
    if (n_stages > 1) rotate_issue(1, rt);
    __syncthreads();                 // rot[0] complete (and the tail partials of slot 0)
    tail_finish(0);
    pending_tail = -1;
    {
      f32x4 v0, v1;
      blend(0, 0, v0, v1);
      publish(0, v0, v1);
    }
    if (n_stages > 1) rotate_commit(1, 1, rt);
    if (n_stages > 2) rotate_issue(2, rt);
    for (int s = 0; s < n_stages; ++s) {
      const int k = 1 + s / NPH, ph = s % NPH;
      __syncthreads();               // frag[s & 1] and rot[(s + 1) & 1] complete; blend(s), mfma(s - 1) done everywhere
      if (pending_tail >= 0) { tail_finish(pending_tail); pending_tail = -1; }
      half8 wh[4], wl[4];            // this stage's weight fragments (L2): in flight under the blend
#pragma unroll
      for (int u = 0; u < 4; ++u) { wh[u] = w1frag(1, ph * 4 + u, 0); wl[u] = w1frag(1, ph * 4 + u, 1); }
      if (s + 1 < n_stages) {
        f32x4 v0, v1;
        blend(s + 1, (s + 1) & 1, v0, v1);
        publish((s + 1) & 1, v0, v1);
      }
      if (ph == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const half8 fh = *reinterpret_cast<const half8*>(&frag_s[s & 1][u][0][lane][0]);
        const half8 fl = *reinterpret_cast<const half8*>(&frag_s[s & 1][u][1][lane][0]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[u], fh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[u], fl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[u], fh, acc, 0, 0, 0);
      }
      if (s + 2 < n_stages) rotate_commit(s + 2, s & 1, rt);     // rot[s & 1] was last read by blend(s), a barrier ago
      if (s + 3 < n_stages) rotate_issue(s + 3, rt);
      if (ph == NPH - 1) {             // the slot's layer 1 is complete: leave the layer-2 partial for the next barrier
        if (NPH == 1) __syncthreads(); // one-stage slots: the previous slot's partials may still be being read
        f32x16 sum;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[r] = accE[r] + acc[r];
        tail_partial(sum);
        pending_tail = k;
      }
    }
  }
  __syncthreads();
  if (pending_tail >= 0) tail_finish(pending_tail);
  __syncthreads();                     // ek_s complete

  float den = 0.f;
  for (int k = 0; k < n; ++k) den += ek_s[k][li];

  // ---- pass 2: fused = sum_k w_k * y_k in list order; this lane keeps channels 16 (4 ph + wave) + 8 lh + 0..7
  f32x4 f0[NPH], f1[NPH];
  {
    const float w0 = ek_s[0][li] / den;
    if (a.weights_out && pvalid && lh == 0 && wave == 0)
      a.weights_out[(((size_t)b * a.ego_count + il) * a.agents + 0) * hw + p] = w0;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      const int ks = ph * 4 + wave;
      f0[ph] = *reinterpret_cast<const f32x4*>(xrow + 16 * ks) * w0;
      f1[ph] = *reinterpret_cast<const f32x4*>(xrow + 16 * ks + 4) * w0;
    }
  }
  if (n_stages > 0) {
    rotate_issue(0, rt);
    rotate_commit(0, 0, rt);
    if (n_stages > 1) rotate_issue(1, rt);
    for (int k = 1; k < n; ++k) {
      const float wk = ek_s[k][li] / den;
      if (a.weights_out && pvalid && lh == 0 && wave == 0)
        a.weights_out[(((size_t)b * a.ego_count + il) * a.agents + k) * hw + p] = wk;
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph) {
        const int s = (k - 1) * NPH + ph;
        __syncthreads();               // rot[s & 1] complete; blend(s - 1) done everywhere
        f32x4 v0, v1;
        blend(s, s & 1, v0, v1);
        f0[ph] += v0 * wk;
        f1[ph] += v1 * wk;
        if (s + 1 < n_stages) rotate_commit(s + 1, (s + 1) & 1, rt);   // last read by blend(s - 1), a barrier ago
        if (s + 2 < n_stages) rotate_issue(s + 2, rt);
      }
    }
  }
#pragma unroll
  for (int ph = 0; ph < NPH; ++ph) store_piece(ph * 4 + wave, f0[ph], f1[ph]);
  note_range(amax);
}

}  // namespace

namespace dn { unsigned range_flags_fuse_warp(bool reset) { return sp_range_flags_here(reset); } }

extern "C" int dn_disco_fuse_warp(const float* feat, const float* trans, const int32_t* num_agent,
                                  const dn_fuse_mlp_params* p, int batch, int agents, int h, int w, int c,
                                  int only_v2i, int ego_first, int ego_count, void* fused_sp, float* fused_nhwc,
                                  float* weights_out, void* stream) {
  DN_REQUIRE(feat && trans && num_agent && p && (fused_sp || fused_nhwc), "fuse_warp: null pointer");
  DN_REQUIRE(batch > 0 && agents > 0 && h > 0 && w > 0, "fuse_warp: empty problem");
  DN_REQUIRE(dn_fuse_mlp_supported(c), "fuse_warp: %d channels unsupported (64, 128 or 256)", c);
  DN_REQUIRE(ego_first >= 0 && ego_count > 0 && ego_first + ego_count <= agents,
             "fuse_warp: ego range [%d, %d) outside 0..%d", ego_first, ego_first + ego_count, agents);
  DN_REQUIRE(agents <= FW_MAX_AGENTS, "fuse_warp: at most %d agents supported (got %d)", FW_MAX_AGENTS, agents);
  DN_REQUIRE((size_t)h * w * c * 4 < (1ull << 31), "fuse_warp: one map must stay below 2 GiB");
  DN_REQUIRE(p->packed && p->s1 && p->t1 && p->s2 && p->t2 && p->s3 && p->t3 && p->w4 && p->b4,
             "fuse_warp: null MLP parameter");
  auto aligned16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  DN_REQUIRE(aligned16(feat) && aligned16(p->packed) && aligned16(fused_sp) && aligned16(fused_nhwc) &&
                 aligned16(p->s1) && aligned16(p->t1) && aligned16(p->s2) && aligned16(p->t2) && aligned16(p->s3) &&
                 aligned16(p->t3) && aligned16(p->w4),
             "fuse_warp: buffers must be 16-byte aligned");
  FuseWarpArgs a;
  a.feat = feat; a.trans = trans; a.num_agent = num_agent;
  a.w1 = (const unsigned char*)p->packed;
  a.w2 = a.w1 + (size_t)2 * 4 * (c / 16) * 2 * 2 * 32 * 16;
  a.s1 = p->s1; a.t1 = p->t1; a.s2 = p->s2; a.t2 = p->t2; a.s3 = p->s3; a.t3 = p->t3; a.w4 = p->w4; a.b4 = p->b4;
  a.fused_sp = (unsigned char*)fused_sp; a.fused_nhwc = fused_nhwc; a.weights_out = weights_out;
  a.batch = batch; a.agents = agents; a.h = h; a.w = w; a.only_v2i = only_v2i;
  a.ego_first = ego_first; a.ego_count = ego_count;
  a.tiles_x = (w + FT_W - 1) / FT_W;
  a.tiles = a.tiles_x * ((h + FT_H - 1) / FT_H);
  dim3 grid(batch * ego_count * a.tiles);
  hipStream_t s = (hipStream_t)stream;
  if (c == 256) hipLaunchKernelGGL(disco_fuse_warp_kernel<256>, grid, dim3(256), 0, s, a);
  else if (c == 128) hipLaunchKernelGGL(disco_fuse_warp_kernel<128>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(disco_fuse_warp_kernel<64>, grid, dim3(256), 0, s, a);
  return dn::check_launch("disco_fuse_warp_kernel");
}
