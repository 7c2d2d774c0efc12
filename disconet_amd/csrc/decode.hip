// Detection decode: foreground probability (2-class softmax) and anchor-relative
// box decode for every anchor of every BEV cell, one coalesced pass (HBM-bound:
// 32 B in, 28 B out per anchor).
//
// Replaces the dense part of upstream:coperception/utils/postprocess.py
// (softmax + box decode vs anchors) that CoDetModule.predict_all runs on the CPU
// after the forward; NMS and mAP stay CPU-side as in the
// reference.
#include "dn_internal.h"

namespace {

__global__ void decode_kernel(const float* __restrict__ cls, const float* __restrict__ loc,
                              const float* __restrict__ anchors, long per_image, long total,
                              float* __restrict__ scores, float* __restrict__ boxes) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const float c0 = cls[2 * i], c1 = cls[2 * i + 1];
    // softmax over {background, foreground}, max-shifted like F.softmax
    const float m = fmaxf(c0, c1);
    const float e0 = expf(c0 - m), e1 = expf(c1 - m);
    scores[i] = e1 / (e0 + e1);
    const float* a = anchors + 6 * (i % per_image);
    const float* t = loc + 6 * i;
    float* o = boxes + 6 * i;
    o[0] = a[0] + t[0] * a[2];
    o[1] = a[1] + t[1] * a[3];
    o[2] = a[2] * expf(t[2]);
    o[3] = a[3] * expf(t[3]);
    o[4] = a[4] * t[5] + a[5] * t[4];
    o[5] = a[5] * t[5] - a[4] * t[4];
  }
}

}  // namespace

extern "C" int dn_decode_boxes(const float* cls, const float* loc, const float* anchors,
                               int n_images, long anchors_per_image, float* scores, float* boxes,
                               void* stream) {
  DN_REQUIRE(cls && loc && anchors && scores && boxes, "decode: null pointer");
  DN_REQUIRE(n_images > 0 && anchors_per_image > 0, "decode: empty problem");
  const long total = (long)n_images * anchors_per_image;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(decode_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, cls, loc,
                     anchors, anchors_per_image, total, scores, boxes);
  return dn::check_launch("decode_kernel");
}
