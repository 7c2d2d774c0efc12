// Device helpers of the pose warp (upstream:coperception/models/det/base/* :: feature_transformation, SURVEY.md
// Appx A.4) of warp.hip: grid_sample's bilinear tap set and a branch-free bilinear read
// of an NHWC fp32 image through a buffer resource.
#pragma once
#include <hip/hip_runtime.h>

#ifndef DN_F32X4_DEFINED
#define DN_F32X4_DEFINED
typedef float f32x4 __attribute__((ext_vector_type(4)));
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

}  // namespace
