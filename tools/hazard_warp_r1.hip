// The pose-warp kernel as it was BEFORE commit 0f7f408 (round 1), kept as a reproducer for
// its channel loop `for (c4 = lane; c4 < c4n; c4 += 64)` has a lane-dependent exit.
// tools/hazard_corun.py runs it beside the conv engine's split-f16 tiles on a second stream and
// counts launches whose output differs from the serial result, for
//   HZ_VARIANT 0  the round-1 loop          1  the shipped fix (wave-uniform trip count)
//   HZ_VARIANT 2  round-1 loop + 2 x s_nop 15 at the loop top
//   HZ_VARIANT 3  round-1 loop + s_waitcnt vmcnt(0) lgkmcnt(0) at the loop bottom
//   HZ_VARIANT 4  round-1 loop + a counter of iterations executed with c4 >= c4n (must stay 0)
//   HZ_VARIANT 6  round-1 loop, float divisions replaced by reciprocal multiplies (no v_div_fmas in the kernel)
//   HZ_VARIANT 7  shipped loop form + division-free coordinates
//   HZ_VARIANT 14 round-1 loop; after it every lane stores its own copy of the wave-uniform tap masks / indices / weights
//   HZ_VARIANT 5  round-1 loop + probes inside it: EXEC not full / lanes that disagree on the wave-uniform
//                 pixel index, tap coordinates or interpolation weights (trace[1..4])
// Build (one shared object per variant):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DHZ_VARIANT=0 tools/hazard_warp_r1.hip -o tools/hazard_warp_v0.so
#include <hip/hip_runtime.h>
#include <cstdint>
#ifndef HZ_VARIANT
#define HZ_VARIANT 0
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

// HZ_VARIANT 6: the round-1 loop, but every float division of the coordinate math is a multiplication by a
// reciprocal computed once (no v_div_scale -> VCC -> v_div_fmas sequence anywhere in the kernel).
// HZ_VARIANT 7: the shipped loop form (as 1) AND division-free coordinates.
#if HZ_VARIANT == 6 || HZ_VARIANT == 7
#define HZ_DIV(a, b) ((a) * __builtin_amdgcn_rcpf((float)(b)))
#elif HZ_VARIANT >= 8
// Variants 8..13: the IEEE division written out with the builtins LLVM's own fdiv lowering uses
// (v_div_scale x2, v_rcp, 5 FMAs, v_div_fmas, v_div_fixup), one ingredient changed at a time:
//    8  the full sequence (expected to fail like the compiler's own)
//    9  v_div_fmas replaced by a plain v_fma (exact when no scaling is needed: always, here)
//   10  v_div_fmas with a constant-false scale flag (s_mov vcc, 0 instead of v_div_scale's VCC)
//   11  full sequence + s_nop 15 x2 before v_div_fmas
//   12  full sequence without v_div_fixup
//   13  full sequence with the numerator-side v_div_scale replaced by a copy (VCC never written by VALU)
__device__ inline float hz_div(float a, float b) {
  bool f0, f1;
  const float ds = __builtin_amdgcn_div_scalef(a, b, false, &f0);
#if HZ_VARIANT == 13
  const float ns = a; f1 = false;
#else
  const float ns = __builtin_amdgcn_div_scalef(a, b, true, &f1);
#endif
  float r = __builtin_amdgcn_rcpf(ds);
  const float e = __builtin_fmaf(-ds, r, 1.f);
  r = __builtin_fmaf(e, r, r);
  float q = ns * r;
  float t = __builtin_fmaf(-ds, q, ns);
  q = __builtin_fmaf(t, r, q);
  t = __builtin_fmaf(-ds, q, ns);
#if HZ_VARIANT == 9
  const float res = __builtin_fmaf(t, r, q);
#elif HZ_VARIANT == 10
  const float res = __builtin_amdgcn_div_fmasf(t, r, q, false);
#else
#if HZ_VARIANT == 11
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#endif
  const float res = __builtin_amdgcn_div_fmasf(t, r, q, f1);
#endif
#if HZ_VARIANT == 12
  return res;
#else
  return __builtin_amdgcn_div_fixupf(res, b, a);
#endif
}
#define HZ_DIV(a, b) hz_div((a), (float)(b))
#else
#define HZ_DIV(a, b) ((a) / (b))
#endif

namespace {

struct Bilinear {
  int x0, y0;          // north-west integer tap
  float w_nw, w_ne, w_sw, w_se;
};

// grid_sample(bilinear, align_corners=False) tap set for normalised (gx, gy)
__device__ inline Bilinear bilinear_taps(float gx, float gy, int w, int h) {
  const float ix = ((gx + 1.f) * w - 1.f) * 0.5f;
  const float iy = ((gy + 1.f) * h - 1.f) * 0.5f;
  const float fx = floorf(ix), fy = floorf(iy);
  Bilinear b;
  b.x0 = (int)fx;
  b.y0 = (int)fy;
  const float ex = fx + 1.f, ey = fy + 1.f;  // south-east corner
  b.w_nw = (ex - ix) * (ey - iy);
  b.w_ne = (ix - fx) * (ey - iy);
  b.w_sw = (ex - ix) * (iy - fy);
  b.w_se = (ix - fx) * (iy - fy);
  return b;
}

__device__ inline f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// One source image as a buffer resource: every tap load is `buffer_load_dwordx4` with a 32-bit
// byte offset, like the conv engine's operand loads (no 64-bit address arithmetic per tap).
struct SrcImage {
  __amdgpu_buffer_rsrc_t rsrc;
};
__device__ inline SrcImage make_src_image(const float* base, size_t bytes) {
  return SrcImage{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000)};
}
__device__ inline f32x4 ldb4(const SrcImage& s, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s.rsrc, byte_off, 0, 0));
}

// bilinear sample of src (one image, [h][w][c]) at taps b, channels c4*4..+3.
// Branch-free: every tap is loaded from a clamped in-frame address and its weight
// is zeroed when the tap is outside, so the 4 loads (16 per output pixel) issue
// back to back instead of draining vmcnt at every divergent join.  A zero weight
// times a finite in-frame value is exactly the zero padding.
__device__ inline f32x4 sample_src(const SrcImage& src, const Bilinear& b, int w, int h, int c,
                                   int c4) {
  const bool x0ok = b.x0 >= 0 && b.x0 < w, x1ok = b.x0 + 1 >= 0 && b.x0 + 1 < w;
  const bool y0ok = b.y0 >= 0 && b.y0 < h, y1ok = b.y0 + 1 >= 0 && b.y0 + 1 < h;
  const int x0 = min(max(b.x0, 0), w - 1), x1 = min(max(b.x0 + 1, 0), w - 1);
  const int y0 = min(max(b.y0, 0), h - 1), y1 = min(max(b.y0 + 1, 0), h - 1);
  const unsigned lane_off = 16u * c4;
  const f32x4 v_nw = ldb4(src, (unsigned)((y0 * w + x0) * c) * 4u + lane_off);
  const f32x4 v_ne = ldb4(src, (unsigned)((y0 * w + x1) * c) * 4u + lane_off);
  const f32x4 v_sw = ldb4(src, (unsigned)((y1 * w + x0) * c) * 4u + lane_off);
  const f32x4 v_se = ldb4(src, (unsigned)((y1 * w + x1) * c) * 4u + lane_off);
  // order matches torch's CPU kernel: nw, ne, sw, se
  f32x4 acc = v_nw * ((y0ok && x0ok) ? b.w_nw : 0.f);
  acc += v_ne * ((y0ok && x1ok) ? b.w_ne : 0.f);
  acc += v_sw * ((y1ok && x0ok) ? b.w_sw : 0.f);
  acc += v_se * ((y1ok && x1ok) ? b.w_se : 0.f);
  return acc;
}

constexpr int PIX_PER_BLOCK = 32;

__global__ void __launch_bounds__(256)
warp_neighbors_kernel(const float* __restrict__ feat, const float* __restrict__ trans,
                      const int32_t* __restrict__ num_agent, int batch, int agents, int h, int w,
                      int c, int only_v2i, int ego_first, int ego_count,
                      float* __restrict__ warped, unsigned* __restrict__ trace, unsigned* __restrict__ dbg) {
  const int jj = blockIdx.y;              // neighbour slot 0..A-2
  const int bi = blockIdx.z;              // b * ego_count + (i - ego_first)
  const int b = bi / ego_count, i = ego_first + bi % ego_count;
  const int j = jj + (jj >= i ? 1 : 0);
  const int n_live = num_agent[b];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4n = c >> 2;
  const int hw = h * w;
  float* dst = warped + ((size_t)bi * (agents - 1) + jj) * hw * c;

  const bool live = i < n_live && j < n_live && !(only_v2i && i != 0 && j != 0);
  const SrcImage src = make_src_image(feat + ((size_t)j * batch + b) * hw * c, (size_t)hw * c * 4);   // image j*B+b

  const float* m = trans + (((size_t)b * agents + i) * agents + j) * 16;
  const float r00 = m[0], r01 = m[1], r10 = m[4], r11 = m[5];
  const float x_trans = (4.f * m[3]) * (1.f / 128.f);
  const float y_trans = -(4.f * m[7]) * (1.f / 128.f);

  for (int pp = wave; pp < PIX_PER_BLOCK; pp += 4) {
    const int p = blockIdx.x * PIX_PER_BLOCK + pp;
    if (p >= hw) break;
    const int py = p / w, px = p % w;
    float* out = dst + (size_t)p * c;
    if (!live) {
      for (int c4 = lane; c4 < c4n; c4 += 64)
        *reinterpret_cast<f32x4*>(out + 4 * c4) = f32x4{0.f, 0.f, 0.f, 0.f};
      continue;
    }
    // pass 2 (translation): normalised base coords of pixel centres
    const float bx = HZ_DIV(2.f * px + 1.f, w) - 1.f;
    const float by = HZ_DIV(2.f * py + 1.f, h) - 1.f;
    const Bilinear t2 = bilinear_taps(bx + x_trans, by + y_trans, w, h);
    // the four rotated-map pixels q this output reads, and their source taps
    Bilinear t1[4];
    bool qok[4];
    float qw[4] = {t2.w_nw, t2.w_ne, t2.w_sw, t2.w_se};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int qx = t2.x0 + (k & 1), qy = t2.y0 + (k >> 1);
      qok[k] = qx >= 0 && qx < w && qy >= 0 && qy < h;
      const float qbx = HZ_DIV(2.f * qx + 1.f, w) - 1.f;
      const float qby = HZ_DIV(2.f * qy + 1.f, h) - 1.f;
      t1[k] = bilinear_taps(r00 * qbx + r01 * qby, r10 * qbx + r11 * qby, w, h);
    }
#if HZ_VARIANT == 1 || HZ_VARIANT == 7      // the shipped fix: wave-uniform trip count, no per-lane condition for whole rows
    for (int it = 0; it < ((c4n + 63) >> 6); ++it) {
      const int c4 = lane + 64 * it;
      if ((c4n & 63) != 0 && c4 >= c4n) continue;
#else                    // the round-1 form: lane-dependent exit (v_cmp -> vcc -> s_andn2 exec)
    for (int c4 = lane; c4 < c4n; c4 += 64) {
#endif
#if HZ_VARIANT == 2
      asm volatile("s_nop 15\n\ts_nop 15");
#endif
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) acc += sample_src(src, t1[k], w, h, c, c4) * (qok[k] ? qw[k] : 0.f);
      *reinterpret_cast<f32x4*>(out + 4 * c4) = acc;
#if HZ_VARIANT == 3
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
#if HZ_VARIANT == 4      // does the lane re-enter with a stale mask?  record every (pixel, c4) a lane executes
      atomicAdd(reinterpret_cast<unsigned*>(trace) + (c4 < c4n ? 0 : 1), 1u);
#endif
#if HZ_VARIANT == 5      // probes: is EXEC full inside the loop (c = 256: every lane has work), and do all
      {                  // lanes agree on the wave-uniform pixel index and tap coordinates?
        const unsigned long long ex = __builtin_amdgcn_read_exec();
        if (ex != ~0ull) atomicAdd(trace + 1, 1u);
        if (p != __builtin_amdgcn_readfirstlane(p)) atomicAdd(trace + 2, 1u);
        if (t1[0].x0 != __builtin_amdgcn_readfirstlane(t1[0].x0) || t1[3].y0 != __builtin_amdgcn_readfirstlane(t1[3].y0))
          atomicAdd(trace + 3, 1u);
        if (__builtin_bit_cast(int, qw[1]) != __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, qw[1])))
          atomicAdd(trace + 4, 1u);
      }
#endif
    }
#if HZ_VARIANT == 14     // round-1 loop, then every lane stores what IT holds for the wave-uniform quantities
    {
      uint4 d;
      d.x = (qok[0] ? 1u : 0u) | (qok[1] ? 2u : 0u) | (qok[2] ? 4u : 0u) | (qok[3] ? 8u : 0u);
      int hsh = 0;
      float ws = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hsh = hsh * 961 + t1[k].x0 * 31 + t1[k].y0;
        ws += (k + 1) * (t1[k].w_nw + 2.f * t1[k].w_ne + 3.f * t1[k].w_sw + 5.f * t1[k].w_se);
      }
      d.y = (unsigned)hsh;
      d.z = __builtin_bit_cast(unsigned, qw[0] + 2.f * qw[1] + 3.f * qw[2] + 5.f * qw[3]);
      d.w = __builtin_bit_cast(unsigned, ws);
      reinterpret_cast<uint4*>(dbg)[(((size_t)bi * (agents - 1) + jj) * hw + p) * 64 + lane] = d;
    }
#endif
  }
}

}  // namespace

extern "C" int hz_warp_neighbors(const float* feat, const float* trans, const int32_t* num_agent, int batch,
                                 int agents, int h, int w, int c, float* warped, unsigned* trace, void* stream, unsigned* dbg) {
  dim3 grid((h * w + PIX_PER_BLOCK - 1) / PIX_PER_BLOCK, agents - 1, batch * agents);
  hipLaunchKernelGGL(warp_neighbors_kernel, grid, dim3(256), 0, (hipStream_t)stream, feat, trans, num_agent, batch,
                     agents, h, w, c, 0, 0, agents, warped, trace, dbg);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
