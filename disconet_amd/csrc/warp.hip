// K4: pose-based two-pass bilinear warp of neighbour feature maps into the ego
// frame (rotate -> zero-pad -> translate).  HBM/L2-bound gather.
//
// Layout: NHWC, so the C channels of one pixel are contiguous; one wavefront
// handles one output pixel at a time with each lane carrying a float4 of
// channels (C = 256 -> exactly one 1 KiB coalesced row per tap).  All tap
// coordinates of a pixel are wave-uniform, so the bounds tests are scalar
// branches, not divergent lanes.
//
// The two resampling passes cannot be merged algebraically (the rotated map is
// zero-padded before it is translated; SURVEY.md Appx A.4), but they can be
// fused in one launch: out2(p) reads <= 4 integer pixels q of the rotated map,
// and each in-frame out1(q) is re-derived from <= 4 source taps with exactly
// the per-element arithmetic of the two-pass form (the intermediate map is
// never written to HBM; the 4x re-derivation is L2-resident work).
//
// Replaces upstream:coperception/models/det/base/* :: feature_transformation
// (SURVEY.md §8 a5).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "disconet_train.h"
#include "dn_internal.h"

#include "warp_device.h"

namespace {

constexpr int PIX_PER_BLOCK = 32;

__global__ void __launch_bounds__(256)
warp_neighbors_kernel(const float* __restrict__ feat, const float* __restrict__ trans,
                      const int32_t* __restrict__ num_agent, int batch, int agents, int h, int w,
                      int c, int only_v2i, int ego_first, int ego_count,
                      float* __restrict__ warped) {
  const int jj = blockIdx.y;              // neighbour slot 0..A-2
  const int bi = blockIdx.z;              // b * ego_count + (i - ego_first)
  const int b = bi / ego_count, i = ego_first + bi % ego_count;
  const int j = jj + (jj >= i ? 1 : 0);
  int n_live = num_agent[b];
  n_live = n_live < 0 ? 0 : (n_live > agents ? agents : n_live);     // a bad count never indexes past the agents
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4n = c >> 2;
  const int hw = h * w;
  float* dst = warped + ((size_t)bi * (agents - 1) + jj) * hw * c;

  const bool live = i < n_live && j < n_live && !(only_v2i && i != 0 && j != 0);
  const SrcImage src = make_src_image(feat + ((size_t)j * batch + b) * hw * c, (size_t)hw * c * 4);   // image j*B+b

  const float* m = trans + (((size_t)b * agents + i) * agents + j) * 16;
  const float r00 = m[0], r01 = m[1], r10 = m[4], r11 = m[5];
  const float x_trans = (4.f * m[3]) / 128.f;
  const float y_trans = -(4.f * m[7]) / 128.f;

  // Channel loop with a wave-uniform trip count; whole rows of 64 lanes (c % 256 == 0) carry no per-lane
  // condition.  (Round 1 believed a lane-dependent exit here was behind the corruption seen beside the conv
  // engine on a second stream; profiles/r02_hazard_repro.txt shows this form fails the same way -- the loop
  // form is a convenience, not a fix.)
  const int n_it = (c4n + 63) >> 6;
  const bool full = (c4n & 63) == 0;
  for (int pp = wave; pp < PIX_PER_BLOCK; pp += 4) {
    const int p = blockIdx.x * PIX_PER_BLOCK + pp;
    if (p >= hw) break;
    const int py = p / w, px = p % w;
    float* out = dst + (size_t)p * c;
    if (!live) {
      for (int it = 0; it < n_it; ++it) {
        const int c4 = lane + 64 * it;
        if (full) *reinterpret_cast<f32x4*>(out + 4 * c4) = f32x4{0.f, 0.f, 0.f, 0.f};
        else if (c4 < c4n) *reinterpret_cast<f32x4*>(out + 4 * c4) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      continue;
    }
    // pass 2 (translation): normalised base coords of pixel centres
    const float bx = (2.f * px + 1.f) / w - 1.f;
    const float by = (2.f * py + 1.f) / h - 1.f;
    const Bilinear t2 = bilinear_taps(bx + x_trans, by + y_trans, w, h);
    // the four rotated-map pixels q this output reads, and their source taps
    Bilinear t1[4];
    bool qok[4];
    float qw[4] = {t2.w_nw, t2.w_ne, t2.w_sw, t2.w_se};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int qx = t2.x0 + (k & 1), qy = t2.y0 + (k >> 1);
      qok[k] = qx >= 0 && qx < w && qy >= 0 && qy < h;
      const float qbx = (2.f * qx + 1.f) / w - 1.f;
      const float qby = (2.f * qy + 1.f) / h - 1.f;
      t1[k] = bilinear_taps(r00 * qbx + r01 * qby, r10 * qbx + r11 * qby, w, h);
    }
    auto channels = [&](int c4) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) acc += sample_src(src, t1[k], w, h, c, c4) * (qok[k] ? qw[k] : 0.f);
      *reinterpret_cast<f32x4*>(out + 4 * c4) = acc;
    };
    if (full) {
      for (int it = 0; it < n_it; ++it) channels(lane + 64 * it);
    } else {
      for (int it = 0; it < n_it; ++it)
        if (lane + 64 * it < c4n) channels(lane + 64 * it);
    }
  }
}

// ---- training form: an explicit list of warps, and the backward scatter ------------------
struct PoseTerms {
  float r00, r01, r10, r11, x_trans, y_trans;
};
__device__ inline PoseTerms pose_terms(const float* m) {
  return PoseTerms{m[0], m[1], m[4], m[5], (4.f * m[3]) / 128.f, -(4.f * m[7]) / 128.f};
}
__device__ inline Bilinear pass2_taps(const PoseTerms& t, int px, int py, int w, int h) {
  const float bx = (2.f * px + 1.f) / w - 1.f;
  const float by = (2.f * py + 1.f) / h - 1.f;
  return bilinear_taps(bx + t.x_trans, by + t.y_trans, w, h);
}
__device__ inline Bilinear pass1_taps(const PoseTerms& t, int qx, int qy, int w, int h) {
  const float qbx = (2.f * qx + 1.f) / w - 1.f;
  const float qby = (2.f * qy + 1.f) / h - 1.f;
  return bilinear_taps(t.r00 * qbx + t.r01 * qby, t.r10 * qbx + t.r11 * qby, w, h);
}

__global__ void __launch_bounds__(256)
warp_list_kernel(const float* __restrict__ src_maps, const float* __restrict__ poses,
                 const int32_t* __restrict__ src_image, int h, int w, int c,
                 float* __restrict__ warped) {
  const int wi = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4n = c >> 2, hw = h * w;
  const SrcImage src = make_src_image(src_maps + (size_t)src_image[wi] * hw * c, (size_t)hw * c * 4);
  const PoseTerms t = pose_terms(poses + 16 * (size_t)wi);
  for (int pp = wave; pp < PIX_PER_BLOCK; pp += 4) {
    const int p = blockIdx.x * PIX_PER_BLOCK + pp;
    if (p >= hw) break;
    const Bilinear t2 = pass2_taps(t, p % w, p / w, w, h);
    Bilinear t1[4];
    bool qok[4];
    float qw[4] = {t2.w_nw, t2.w_ne, t2.w_sw, t2.w_se};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int qx = t2.x0 + (k & 1), qy = t2.y0 + (k >> 1);
      qok[k] = qx >= 0 && qx < w && qy >= 0 && qy < h;
      t1[k] = pass1_taps(t, qx, qy, w, h);
    }
    float* out = warped + ((size_t)wi * hw + p) * c;
    for (int c4 = lane; c4 < c4n; c4 += 64) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) acc += sample_src(src, t1[k], w, h, c, c4) * (qok[k] ? qw[k] : 0.f);
      *reinterpret_cast<f32x4*>(out + 4 * c4) = acc;
    }
  }
}

// Transposes of the two gathers, in reverse order.  PASS = 2: gradient of the translated map
// -> gradient of the rotated map (scratch, per warp).  PASS = 1: rotated-map gradient ->
// the source agent's map.  Several outputs share a tap: hardware float atomics (L2).
template <int PASS>
__global__ void __launch_bounds__(256)
warp_scatter_kernel(const float* __restrict__ grad_in, const float* __restrict__ poses,
                    const int32_t* __restrict__ src_image, int h, int w, int c,
                    float* __restrict__ grad_out) {
  const int wi = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4n = c >> 2, hw = h * w;
  const PoseTerms t = pose_terms(poses + 16 * (size_t)wi);
  float* dst = grad_out + (size_t)(PASS == 2 ? wi : src_image[wi]) * hw * c;
  for (int pp = wave; pp < PIX_PER_BLOCK; pp += 4) {
    const int p = blockIdx.x * PIX_PER_BLOCK + pp;
    if (p >= hw) break;
    const Bilinear b = PASS == 2 ? pass2_taps(t, p % w, p / w, w, h) : pass1_taps(t, p % w, p / w, w, h);
    const float wt[4] = {b.w_nw, b.w_ne, b.w_sw, b.w_se};
    const float* gin = grad_in + ((size_t)wi * hw + p) * c;
    for (int c4 = lane; c4 < c4n; c4 += 64) {
      const f32x4 g = ld4(gin + 4 * c4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int x = b.x0 + (k & 1), y = b.y0 + (k >> 1);
        if (x < 0 || x >= w || y < 0 || y >= h) continue;   // wave-uniform
        float* o = dst + ((size_t)y * w + x) * c + 4 * c4;
#pragma unroll
        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(o + e, g[e] * wt[k]);
      }
    }
  }
}

// Gather forms of the same two transposes for RIGID poses (rotation + translation) on square
// maps: deterministic, no atomics, no zero-filled scratch.  Every output pixel looks at the few
// input pixels whose bilinear footprint can reach it and re-derives their taps with exactly the
// forward's arithmetic, so the result is the scatter's sum in a fixed order.
//   pass 2 (translation): out1_grad[q] = sum over p with q in taps2(p): the taps of p sit at
//     p + floor(shift) (+1), so p lies within [-2, +1] of q - floor(shift) per axis (float fuzz incl.);
//   pass 1 (rotation):    src_grad[r] += sum over warps w of source image m, over q with r in
//     taps1(q): for an orthonormal R the footprint of q reaches r only if q is within sqrt(2) of
//     R^-1 r, i.e. within +-2 of its rounding.
__global__ void __launch_bounds__(256)
warp_gather2_kernel(const float* __restrict__ d_warped, const float* __restrict__ poses, int h, int w,
                    int c, float* __restrict__ d_out1) {
  const int wi = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4n = c >> 2, hw = h * w;
  const PoseTerms t = pose_terms(poses + 16 * (size_t)wi);
  const int fsx = (int)floorf(t.x_trans * w * 0.5f), fsy = (int)floorf(t.y_trans * h * 0.5f);
  const float* gin = d_warped + (size_t)wi * hw * c;
  for (int pp = wave; pp < PIX_PER_BLOCK; pp += 4) {
    const int q = blockIdx.x * PIX_PER_BLOCK + pp;
    if (q >= hw) break;
    const int qx = q % w, qy = q / w;
    f32x4 acc[4];                                  // c <= 1024
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int dy = -2; dy <= 1; ++dy) {
      const int py = qy - fsy + dy;
      if (py < 0 || py >= h) continue;
      for (int dx = -2; dx <= 1; ++dx) {
        const int px = qx - fsx + dx;
        if (px < 0 || px >= w) continue;
        const Bilinear b = pass2_taps(t, px, py, w, h);
        const int ox = qx - b.x0, oy = qy - b.y0;
        if (ox < 0 || ox > 1 || oy < 0 || oy > 1) continue;   // wave-uniform
        const float wt = oy ? (ox ? b.w_se : b.w_sw) : (ox ? b.w_ne : b.w_nw);
        const float* g = gin + (size_t)(py * w + px) * c;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (lane + 64 * i < c4n) acc[i] += ld4(g + 4 * (lane + 64 * i)) * wt;
      }
    }
    float* out = d_out1 + ((size_t)wi * hw + q) * c;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (lane + 64 * i < c4n) *reinterpret_cast<f32x4*>(out + 4 * (lane + 64 * i)) = acc[i];
  }
}

__global__ void __launch_bounds__(256)
warp_gather1_kernel(const float* __restrict__ d_out1, const float* __restrict__ poses,
                    const int32_t* __restrict__ src_image, int n_warps, int h, int w, int c,
                    float* __restrict__ d_src) {
  const int m = blockIdx.y;                       // source image
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4n = c >> 2, hw = h * w;
  for (int pp = wave; pp < PIX_PER_BLOCK; pp += 4) {
    const int r = blockIdx.x * PIX_PER_BLOCK + pp;
    if (r >= hw) break;
    const int rx = r % w, ry = r / w;
    const float rbx = (2.f * rx + 1.f) / w - 1.f, rby = (2.f * ry + 1.f) / h - 1.f;
    float* out = d_src + ((size_t)m * hw + r) * c;
    f32x4 acc[4];                                  // c <= 1024
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc[i] = (lane + 64 * i < c4n) ? ld4(out + 4 * (lane + 64 * i)) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int wi = 0; wi < n_warps; ++wi) {
      if (src_image[wi] != m) continue;            // wave-uniform
      const PoseTerms t = pose_terms(poses + 16 * (size_t)wi);
      const float det = t.r00 * t.r11 - t.r01 * t.r10;
      const float qbx = (t.r11 * rbx - t.r01 * rby) / det, qby = (-t.r10 * rbx + t.r00 * rby) / det;
      const int cx = (int)rintf(((qbx + 1.f) * w - 1.f) * 0.5f), cy = (int)rintf(((qby + 1.f) * h - 1.f) * 0.5f);
      const float* gin = d_out1 + (size_t)wi * hw * c;
      for (int dy = -2; dy <= 2; ++dy) {
        const int qy = cy + dy;
        if (qy < 0 || qy >= h) continue;
        for (int dx = -2; dx <= 2; ++dx) {
          const int qx = cx + dx;
          if (qx < 0 || qx >= w) continue;
          const Bilinear b = pass1_taps(t, qx, qy, w, h);
          const int ox = rx - b.x0, oy = ry - b.y0;
          if (ox < 0 || ox > 1 || oy < 0 || oy > 1) continue;
          const float wt = oy ? (ox ? b.w_se : b.w_sw) : (ox ? b.w_ne : b.w_nw);
          const float* g = gin + (size_t)(qy * w + qx) * c;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (lane + 64 * i < c4n) acc[i] += ld4(g + 4 * (lane + 64 * i)) * wt;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (lane + 64 * i < c4n) *reinterpret_cast<f32x4*>(out + 4 * (lane + 64 * i)) = acc[i];
  }
}

// ---- lane-parallel candidate search of the two gathers (round 6).  The kernels above evaluate their 16 / 25 candidate pixels one
// after the other with wave-uniform arithmetic -- 64 lanes computing the same tap set, ~100 evaluations per output pixel in pass 1:
// VALU-bound at 228 + 92 us per step for 32 x 32 x 256 maps.  Here lane k evaluates candidate k (the same expressions: pass1_taps /
// pass2_taps of the same integers), a ballot gives the hits, and the hits are added in ascending candidate order -- the order of the
// loops above -- four rows in flight at a time.  Deterministic like them; against them the sums agree to a few ulp, not bit for bit
// (the compiler contracts the tap arithmetic per kernel: weights differ in their last bit).  PIXW: output pixels per wave.
#ifndef DN_GATHER_PIXW
#define DN_GATHER_PIXW 1
#endif
constexpr int GATHER_PIXW = DN_GATHER_PIXW;
struct GatherHits {
  unsigned long long mask;
  float wt;
  int off;
};
// adds the hit rows (row offset `off`, weight `wt` of the lanes in `mask`, ascending) of `gin` into acc
__device__ inline void gather_add_hits(GatherHits hs, const float* __restrict__ gin, int c, int c4n, int lane, f32x4 (&acc)[4]) {
  while (hs.mask) {
    int n = 0;
    float wk[4];
    int ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (hs.mask) {
        const int kk = __builtin_ctzll(hs.mask);
        hs.mask &= hs.mask - 1;
        wk[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hs.wt), kk));
        ok[j] = __builtin_amdgcn_readlane(hs.off, kk);
        n = j + 1;
      } else {
        wk[j] = 0.f;
        ok[j] = 0;
      }
    }
    f32x4 v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < n) {                                   // wave-uniform
        const float* g = gin + (size_t)ok[j] * c;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (lane + 64 * i < c4n) v[j][i] = ld4(g + 4 * (lane + 64 * i));
      }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < n) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (lane + 64 * i < c4n) acc[i] += v[j][i] * wk[j];
      }
  }
}

__global__ void __launch_bounds__(256)
warp_gather2_lanes_kernel(const float* __restrict__ d_warped, const float* __restrict__ poses, int h, int w,
                          int c, float* __restrict__ d_out1) {
  const int wi = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4n = c >> 2, hw = h * w;
  const PoseTerms t = pose_terms(poses + 16 * (size_t)wi);
  const int fsx = (int)floorf(t.x_trans * w * 0.5f), fsy = (int)floorf(t.y_trans * h * 0.5f);
  const float* gin = d_warped + (size_t)wi * hw * c;
  const int k = lane & 15, dy = (k >> 2) - 2, dx = (k & 3) - 2;      // candidate k of the scalar kernel's (dy, dx) loops
  for (int pp = 0; pp < GATHER_PIXW; ++pp) {
    const int q = (blockIdx.x * 4 + wave) * GATHER_PIXW + pp;
    if (q >= hw) break;
    const int qx = q % w, qy = q / w;
    f32x4 acc[4];                                  // c <= 1024
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int py = qy - fsy + dy, px = qx - fsx + dx;
    bool hit = lane < 16 && py >= 0 && py < h && px >= 0 && px < w;
    const Bilinear b = pass2_taps(t, px, py, w, h);
    const int ox = qx - b.x0, oy = qy - b.y0;
    hit = hit && ox >= 0 && ox <= 1 && oy >= 0 && oy <= 1;
    GatherHits hs;
    hs.wt = oy ? (ox ? b.w_se : b.w_sw) : (ox ? b.w_ne : b.w_nw);
    hs.off = py * w + px;
    hs.mask = __ballot(hit);
    gather_add_hits(hs, gin, c, c4n, lane, acc);
    float* out = d_out1 + ((size_t)wi * hw + q) * c;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (lane + 64 * i < c4n) *reinterpret_cast<f32x4*>(out + 4 * (lane + 64 * i)) = acc[i];
  }
}

__global__ void __launch_bounds__(256)
warp_gather1_lanes_kernel(const float* __restrict__ d_out1, const float* __restrict__ poses,
                          const int32_t* __restrict__ src_image, int n_warps, int h, int w, int c,
                          float* __restrict__ d_src) {
  const int m = blockIdx.y;                       // source image
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4n = c >> 2, hw = h * w;
  const int k = lane < 25 ? lane : 24, dy = k / 5 - 2, dx = k % 5 - 2;   // candidate k of the scalar kernel's (dy, dx) loops
  for (int pp = 0; pp < GATHER_PIXW; ++pp) {
    const int r = (blockIdx.x * 4 + wave) * GATHER_PIXW + pp;
    if (r >= hw) break;
    const int rx = r % w, ry = r / w;
    const float rbx = (2.f * rx + 1.f) / w - 1.f, rby = (2.f * ry + 1.f) / h - 1.f;
    float* out = d_src + ((size_t)m * hw + r) * c;
    f32x4 acc[4];                                  // c <= 1024
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc[i] = (lane + 64 * i < c4n) ? ld4(out + 4 * (lane + 64 * i)) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int w0 = 0; w0 < n_warps; w0 += 64) {
      unsigned long long wmask = __ballot(w0 + lane < n_warps && src_image[w0 + lane < n_warps ? w0 + lane : 0] == m);
      while (wmask) {                              // the warps of this source image, ascending
        const int wi = w0 + __builtin_ctzll(wmask);
        wmask &= wmask - 1;
        const PoseTerms t = pose_terms(poses + 16 * (size_t)wi);
        const float det = t.r00 * t.r11 - t.r01 * t.r10;
        const float qbx = (t.r11 * rbx - t.r01 * rby) / det, qby = (-t.r10 * rbx + t.r00 * rby) / det;
        const int cx = (int)rintf(((qbx + 1.f) * w - 1.f) * 0.5f), cy = (int)rintf(((qby + 1.f) * h - 1.f) * 0.5f);
        const int qy = cy + dy, qx = cx + dx;
        bool hit = lane < 25 && qy >= 0 && qy < h && qx >= 0 && qx < w;
        const Bilinear b = pass1_taps(t, qx, qy, w, h);
        const int ox = rx - b.x0, oy = ry - b.y0;
        hit = hit && ox >= 0 && ox <= 1 && oy >= 0 && oy <= 1;
        GatherHits hs;
        hs.wt = oy ? (ox ? b.w_se : b.w_sw) : (ox ? b.w_ne : b.w_nw);
        hs.off = qy * w + qx;
        hs.mask = __ballot(hit);
        gather_add_hits(hs, d_out1 + (size_t)wi * hw * c, c, c4n, lane, acc);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (lane + 64 * i < c4n) *reinterpret_cast<f32x4*>(out + 4 * (lane + 64 * i)) = acc[i];
  }
}

// ---- tile-shared form of the same two passes.  Every output pixel p of an 8x8 tile reads the four
// rotated-map pixels q = (x0(p) + {0,1}, y0(p) + {0,1}); the translation is one offset per (ego, neighbour)
// pair, so the tile's q pixels form a (T+1)^2 block (T+2 allowed for a rounding split of floor()).  The
// rotated map R(q) of that block is computed ONCE per workgroup into LDS (4 taps of the source map each,
// with sample_src's arithmetic) and the tile's 64 outputs are blended from LDS with the per-pixel weights
// of the one-pixel-per-wave kernel -- the same values in the same order, with ~3x fewer L2 tap reads
// (the one-pixel kernel re-derives every R(q) four times and is bound by L2 -> L1 bandwidth, ~1.3 GB per
// step).  A workgroup covers one 64-channel slice of the tile: 100 x 256 B = 25.6 KB of LDS.
constexpr int WT = 8, WQ = WT + 2, WCS = 64;
typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
typedef int i32x4w __attribute__((ext_vector_type(4)));
#ifndef DN_WARP_SHARED_TAPS
#define DN_WARP_SHARED_TAPS 1   // 0: round 3's form (every lane derives its pixel's tap sets), for A/B builds
#endif

__global__ void __launch_bounds__(256)
warp_neighbors_tiled_kernel(const float* __restrict__ feat, const float* __restrict__ trans,
                            const int32_t* __restrict__ num_agent, int batch, int agents, int h, int w,
                            int c, int only_v2i, int ego_first, int ego_count, int tiles_x, int fm,
                            float* __restrict__ warped) {
  __shared__ f32x4 rot[WQ * WQ][WCS / 4];
  const int n_slices = c / WCS;
  const int slice = blockIdx.x % n_slices, tile = blockIdx.x / n_slices;
  const int tile_x0 = (tile % tiles_x) * WT, tile_y0 = (tile / tiles_x) * WT;
  const int jj = blockIdx.y, bi = blockIdx.z;
  const int b = bi / ego_count, i = ego_first + bi % ego_count;
  const int j = jj + (jj >= i ? 1 : 0);
  int n_live = num_agent[b];
  n_live = n_live < 0 ? 0 : (n_live > agents ? agents : n_live);
  const int tid = threadIdx.x, l = tid & 15;
  const int hw = h * w;
  float* dst = warped + ((size_t)bi * (agents - 1) + jj) * hw * c + (fm ? 0 : slice * WCS + 4 * l);
  // fm: FRAGMENT-MAJOR pair block (include/disconet_hip.h :: dn_warp_neighbors_fm) -- the order in which the attention
  // launch reads it: [tile of 32 pixels][k-step of 16 channels][half r][lane = 32 h + j] x 4 floats, channel
  // 16 ks + 8 h + 4 r + e of pixel 32 t + j.  This thread's four channels 64 slice + 4 l .. + 3 are one such piece.
  const int fm_ks = slice * 4 + (l >> 2), fm_h = (l >> 1) & 1, fm_r = l & 1, fm_kss = c >> 4;
  auto out_of = [&](int p) {
    return fm ? dst + (((((size_t)(p >> 5) * fm_kss + fm_ks) * 2 + fm_r) * 64) + fm_h * 32 + (p & 31)) * 4
              : dst + (size_t)p * c;
  };
  const bool live = i < n_live && j < n_live && !(only_v2i && i != 0 && j != 0);
  if (!live) {
    for (int pl = tid >> 4; pl < WT * WT; pl += 16) {
      const int px = tile_x0 + pl % WT, py = tile_y0 + pl / WT;
      if (px < w && py < h) *reinterpret_cast<f32x4*>(out_of(py * w + px)) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return;
  }
  const SrcImage src = make_src_image(feat + ((size_t)j * batch + b) * hw * c, (size_t)hw * c * 4);
  const float* m = trans + (((size_t)b * agents + i) * agents + j) * 16;
  const float r00 = m[0], r01 = m[1], r10 = m[4], r11 = m[5];
  const float x_trans = (4.f * m[3]) / 128.f;
  const float y_trans = -(4.f * m[7]) / 128.f;
#if DN_WARP_SHARED_TAPS
  // The tap sets depend on the pixel only, and sixteen lanes share a pixel: computed ONCE per pixel into LDS (clamped byte
  // offsets / LDS rows and validity-masked weights, the expressions of sample_src and of the blend below), then every lane
  // does loads and FMAs only.  Round 3's form re-derived them in every lane: 1290 VALU instructions per wave, half of the
  // launch's wave cycles stalled on VALU issue (profiles/r04_fuse_pmc.txt).  Same values, same order of operations.
  __shared__ __attribute__((aligned(16))) unsigned tap1_off[WQ * WQ][4];
  __shared__ __attribute__((aligned(16))) float tap1_w[WQ * WQ][4];
  __shared__ __attribute__((aligned(16))) int tap2_row[WT * WT][4];
  __shared__ __attribute__((aligned(16))) float tap2_w[WT * WT][4];
  // north-west q of the tile: the smallest x0(p) - (p - tile origin) over the tile's columns / rows; column / row k on lane
  // k mod 8, minimum over each group of eight lanes (every group holds all eight)
  int qx_base, qy_base;
  {
    const int k = tid & 7;
    const Bilinear t = bilinear_taps((2.f * (tile_x0 + k) + 1.f) / w - 1.f + x_trans,
                                     (2.f * (tile_y0 + k) + 1.f) / h - 1.f + y_trans, w, h);
    qx_base = t.x0 - k;
    qy_base = t.y0 - k;
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
      qx_base = min(qx_base, __shfl_xor(qx_base, d, 64));
      qy_base = min(qy_base, __shfl_xor(qy_base, d, 64));
    }
  }
  if (tid < WQ * WQ) {              // pass-1 tap set of rotated pixel q = tid
    const int q = tid;
    const int qx = qx_base + q % WQ, qy = qy_base + q / WQ;
    const float qbx = (2.f * qx + 1.f) / w - 1.f;
    const float qby = (2.f * qy + 1.f) / h - 1.f;
    const Bilinear t1 = bilinear_taps(r00 * qbx + r01 * qby, r10 * qbx + r11 * qby, w, h);
    const bool x0ok = t1.x0 >= 0 && t1.x0 < w, x1ok = t1.x0 + 1 >= 0 && t1.x0 + 1 < w;
    const bool y0ok = t1.y0 >= 0 && t1.y0 < h, y1ok = t1.y0 + 1 >= 0 && t1.y0 + 1 < h;
    const int x0 = min(max(t1.x0, 0), w - 1), x1 = min(max(t1.x0 + 1, 0), w - 1);
    const int y0 = min(max(t1.y0, 0), h - 1), y1 = min(max(t1.y0 + 1, 0), h - 1);
    *reinterpret_cast<u32x4w*>(tap1_off[q]) = u32x4w{(unsigned)((y0 * w + x0) * c) * 4u, (unsigned)((y0 * w + x1) * c) * 4u,
                                                      (unsigned)((y1 * w + x0) * c) * 4u, (unsigned)((y1 * w + x1) * c) * 4u};
    *reinterpret_cast<f32x4*>(tap1_w[q]) = f32x4{(y0ok && x0ok) ? t1.w_nw : 0.f, (y0ok && x1ok) ? t1.w_ne : 0.f,
                                                 (y1ok && x0ok) ? t1.w_sw : 0.f, (y1ok && x1ok) ? t1.w_se : 0.f};
  } else if (tid >= 128 && tid < 128 + WT * WT) {     // pass-2 tap set of output pixel pl (a wave of its own)
    const int pl = tid - 128;
    const int px = tile_x0 + pl % WT, py = tile_y0 + pl / WT;
    const float bx = (2.f * px + 1.f) / w - 1.f;
    const float by = (2.f * py + 1.f) / h - 1.f;
    const Bilinear t2 = bilinear_taps(bx + x_trans, by + y_trans, w, h);
    const float qw[4] = {t2.w_nw, t2.w_ne, t2.w_sw, t2.w_se};
    int row[4];
    float wk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int qx = t2.x0 + (k & 1), qy = t2.y0 + (k >> 1);
      const bool qok = qx >= 0 && qx < w && qy >= 0 && qy < h;
      const int lx = min(max(qx - qx_base, 0), WQ - 1), ly = min(max(qy - qy_base, 0), WQ - 1);
      row[k] = ly * WQ + lx;
      wk[k] = qok ? qw[k] : 0.f;
    }
    *reinterpret_cast<i32x4w*>(tap2_row[pl]) = i32x4w{row[0], row[1], row[2], row[3]};
    *reinterpret_cast<f32x4*>(tap2_w[pl]) = f32x4{wk[0], wk[1], wk[2], wk[3]};
  }
  __syncthreads();
  // pass 1 (rotation) of the tile's q block: nw, ne, sw, se as torch's CPU kernel (sample_src)
  const unsigned lane_off = 16u * (slice * (WCS / 4) + l);
  for (int idx = tid; idx < WQ * WQ * 16; idx += 256) {
    const int q = idx >> 4;
    const u32x4w off = *reinterpret_cast<const u32x4w*>(tap1_off[q]);
    const f32x4 wt = *reinterpret_cast<const f32x4*>(tap1_w[q]);
    const f32x4 v_nw = ldb4(src, off[0] + lane_off), v_ne = ldb4(src, off[1] + lane_off);
    const f32x4 v_sw = ldb4(src, off[2] + lane_off), v_se = ldb4(src, off[3] + lane_off);
    f32x4 acc = v_nw * wt[0];
    acc += v_ne * wt[1];
    acc += v_sw * wt[2];
    acc += v_se * wt[3];
    rot[q][l] = acc;
  }
  __syncthreads();
  // pass 2 (translation): blend the four q of every output pixel
  for (int pl = tid >> 4; pl < WT * WT; pl += 16) {
    const int px = tile_x0 + pl % WT, py = tile_y0 + pl / WT;
    if (px >= w || py >= h) continue;
    const i32x4w row = *reinterpret_cast<const i32x4w*>(tap2_row[pl]);
    const f32x4 wt = *reinterpret_cast<const f32x4*>(tap2_w[pl]);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += rot[row[k]][l] * wt[k];
    *reinterpret_cast<f32x4*>(out_of(py * w + px)) = acc;
  }
#else
  // north-west q of the tile: the smallest x0(p) - (p - tile origin) over the tile's columns / rows
  int qx_base = 1 << 30, qy_base = 1 << 30;
#pragma unroll
  for (int k = 0; k < WT; ++k) {
    const Bilinear t = bilinear_taps((2.f * (tile_x0 + k) + 1.f) / w - 1.f + x_trans,
                                     (2.f * (tile_y0 + k) + 1.f) / h - 1.f + y_trans, w, h);
    qx_base = min(qx_base, t.x0 - k);
    qy_base = min(qy_base, t.y0 - k);
  }
  // pass 1 (rotation) of the tile's q block
  for (int idx = tid; idx < WQ * WQ * 16; idx += 256) {
    const int q = idx >> 4;
    const int qx = qx_base + q % WQ, qy = qy_base + q / WQ;
    const float qbx = (2.f * qx + 1.f) / w - 1.f;
    const float qby = (2.f * qy + 1.f) / h - 1.f;
    const Bilinear t1 = bilinear_taps(r00 * qbx + r01 * qby, r10 * qbx + r11 * qby, w, h);
    rot[q][l] = sample_src(src, t1, w, h, c, slice * (WCS / 4) + l);
  }
  __syncthreads();
  // pass 2 (translation): blend the four q of every output pixel
  for (int pl = tid >> 4; pl < WT * WT; pl += 16) {
    const int px = tile_x0 + pl % WT, py = tile_y0 + pl / WT;
    if (px >= w || py >= h) continue;
    const float bx = (2.f * px + 1.f) / w - 1.f;
    const float by = (2.f * py + 1.f) / h - 1.f;
    const Bilinear t2 = bilinear_taps(bx + x_trans, by + y_trans, w, h);
    const float qw[4] = {t2.w_nw, t2.w_ne, t2.w_sw, t2.w_se};
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int qx = t2.x0 + (k & 1), qy = t2.y0 + (k >> 1);
      const bool qok = qx >= 0 && qx < w && qy >= 0 && qy < h;
      const int lx = min(max(qx - qx_base, 0), WQ - 1), ly = min(max(qy - qy_base, 0), WQ - 1);
      acc += rot[ly * WQ + lx][l] * (qok ? qw[k] : 0.f);
    }
    *reinterpret_cast<f32x4*>(out_of(py * w + px)) = acc;
  }
#endif
}

}  // namespace



extern "C" int dn_warp_list(const float* src, const float* poses, const int32_t* src_image,
                            int n_warps, int h, int w, int c, float* warped, void* stream) {
  DN_REQUIRE(src && poses && src_image && warped, "warp list: null pointer");
  DN_REQUIRE(n_warps > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "warp list: bad shape");
  dim3 grid((h * w + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK, n_warps);
  hipLaunchKernelGGL(warp_list_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, poses, src_image, h,
                     w, c, warped);
  return dn::check_launch("warp_list_kernel");
}

extern "C" int dn_warp_backward(const float* d_warped, const float* poses, const int32_t* src_image,
                                int n_warps, int n_src_images, int h, int w, int c, int rigid,
                                float* scratch, float* d_src, void* stream) {
  DN_REQUIRE(d_warped && poses && src_image && scratch && d_src, "warp backward: null pointer");
  DN_REQUIRE(n_warps > 0 && n_src_images > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && c <= 1024,
             "warp backward: bad shape");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((h * w + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK, n_warps);
  if (rigid && h == w) {
    static const bool legacy = [] { const char* e = getenv("DN_WARP_GATHER_LEGACY"); return e && e[0] == '1'; }();
    if (legacy) {     // round 5's one-candidate-at-a-time kernels (tests, A/B runs)
      hipLaunchKernelGGL(warp_gather2_kernel, grid, dim3(256), 0, s, d_warped, poses, h, w, c, scratch);
      hipLaunchKernelGGL(warp_gather1_kernel, dim3(grid.x, n_src_images), dim3(256), 0, s, scratch, poses,
                         src_image, n_warps, h, w, c, d_src);
    } else {
      const int gx = (h * w + 4 * GATHER_PIXW - 1) / (4 * GATHER_PIXW);
      hipLaunchKernelGGL(warp_gather2_lanes_kernel, dim3(gx, n_warps), dim3(256), 0, s, d_warped, poses, h, w, c, scratch);
      hipLaunchKernelGGL(warp_gather1_lanes_kernel, dim3(gx, n_src_images), dim3(256), 0, s, scratch, poses,
                         src_image, n_warps, h, w, c, d_src);
    }
    return dn::check_launch("warp_gather kernels");
  }
  if (dn::zero_fill(scratch, sizeof(float) * (size_t)n_warps * h * w * c, s) != hipSuccess)
    return dn::fail(DN_ERR_LAUNCH, "warp backward: memset failed");
  hipLaunchKernelGGL(warp_scatter_kernel<2>, grid, dim3(256), 0, s, d_warped, poses, src_image, h, w, c,
                     scratch);
  hipLaunchKernelGGL(warp_scatter_kernel<1>, grid, dim3(256), 0, s, scratch, poses, src_image, h, w, c,
                     d_src);
  return dn::check_launch("warp_scatter_kernel");
}

namespace {
int warp_neighbors_impl(const float* feat, const float* trans, const int32_t* num_agent, int batch, int agents, int h, int w,
                        int c, int only_v2i, int ego_first, int ego_count, float* warped, int fm, void* stream);
}

extern "C" int dn_warp_neighbors(const float* feat, const float* trans, const int32_t* num_agent,
                                 int batch, int agents, int h, int w, int c, int only_v2i,
                                 int ego_first, int ego_count, float* warped, void* stream) {
  return warp_neighbors_impl(feat, trans, num_agent, batch, agents, h, w, c, only_v2i, ego_first, ego_count, warped, 0, stream);
}

extern "C" int dn_warp_fm_supported(int h, int w, int c) { return c % WCS == 0 && (h * w) % 32 == 0; }

extern "C" int dn_warp_neighbors_fm(const float* feat, const float* trans, const int32_t* num_agent,
                                    int batch, int agents, int h, int w, int c, int only_v2i,
                                    int ego_first, int ego_count, float* warped, void* stream) {
  DN_REQUIRE(dn_warp_fm_supported(h, w, c), "warp (fragment-major): needs c %% 64 == 0 and h * w %% 32 == 0 (got %d x %d x %d)", h, w, c);
  return warp_neighbors_impl(feat, trans, num_agent, batch, agents, h, w, c, only_v2i, ego_first, ego_count, warped, 1, stream);
}

namespace {
int warp_neighbors_impl(const float* feat, const float* trans, const int32_t* num_agent, int batch, int agents, int h, int w,
                        int c, int only_v2i, int ego_first, int ego_count, float* warped, int fm, void* stream) {
  DN_REQUIRE(batch > 0 && agents > 0 && h > 0 && w > 0, "warp: empty problem");
  DN_REQUIRE(feat && trans && num_agent && (warped || agents < 2), "warp: null pointer");
  DN_REQUIRE(c > 0 && c % 4 == 0, "warp: channel count %d must be a multiple of 4", c);
  DN_REQUIRE(ego_first >= 0 && ego_count > 0 && ego_first + ego_count <= agents,
             "warp: ego range [%d, %d) outside 0..%d", ego_first, ego_first + ego_count, agents);
  if (agents < 2) return DN_OK;   // no neighbours to warp
  const int hw = h * w;
  static const int tiled_env = [] { const char* e = getenv("DN_WARP_TILED"); return e ? atoi(e) : 1; }();
  if ((tiled_env || fm) && c % WCS == 0) {
    const int tiles_x = (w + WT - 1) / WT, tiles_y = (h + WT - 1) / WT;
    dim3 tgrid(tiles_x * tiles_y * (c / WCS), agents - 1, batch * ego_count);
    hipLaunchKernelGGL(warp_neighbors_tiled_kernel, tgrid, dim3(256), 0, (hipStream_t)stream, feat, trans,
                       num_agent, batch, agents, h, w, c, only_v2i, ego_first, ego_count, tiles_x, fm, warped);
    return dn::check_launch("warp_neighbors_tiled_kernel");
  }
  dim3 grid((hw + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK, agents - 1, batch * ego_count);
  hipLaunchKernelGGL(warp_neighbors_kernel, grid, dim3(256), 0, (hipStream_t)stream, feat, trans,
                     num_agent, batch, agents, h, w, c, only_v2i, ego_first, ego_count, warped);
  return dn::check_launch("warp_neighbors_kernel");
}
}  // namespace
