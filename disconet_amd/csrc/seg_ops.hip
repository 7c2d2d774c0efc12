// Segmentation variant of `--com disco` (BASELINE.json configs[3]): the three
// ops the UNet needs beside the conv engine and the fusion block, on split-planar (SP) tensors
// (sp_layout.h) so the maps never leave the conv engine's layout:
//   dn_sp_maxpool2            nn.MaxPool2d(2)                               (Down blocks)
//   dn_sp_upsample2_bilinear  nn.Upsample(x2, bilinear, align_corners=True) (Up blocks)
//   dn_seg_ce_loss            nn.CrossEntropyLoss over the class logits: value and d/d(logits)
// All three are HBM-bound streaming passes: one thread per 16-byte piece pair (8 channels of one
// output pixel), pixel fastest -> every load / store instruction is a contiguous run of a plane.
//
// Replaces the MaxPool2d / Upsample / CrossEntropyLoss calls of
// upstream:coperception/models/seg/SegModelBase.py and upstream:coperception/utils/SegModule.py
// (source not mounted: /root/reference/coperception is an empty submodule directory; the only
// mounted mention of the task is /root/reference/README.md:15).
#include "dn_internal.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

__device__ inline float sp_value(const half8 hi, const half8 lo, int e) { return (float)hi[e] + (float)lo[e]; }

__device__ inline void sp_split(float x, _Float16& hi, _Float16& lo) {
  x = fminf(fmaxf(x, -65504.f), 65504.f);
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}

// idx over (img, chunk, oct, oy, ox) of the OUTPUT, ox fastest
__global__ void sp_maxpool2_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                   int cg_total, int h, int w, long total) {
  const int ho = h >> 1, wo = w >> 1;
  const size_t ip = (size_t)h * w * 16, op = (size_t)ho * wo * 16;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int ox = idx % wo;
    long r = idx / wo;
    const int oy = r % ho; r /= ho;
    const int oct = r % 2; r /= 2;          // r = img * cg_total + cg
    const unsigned char* s = src + ((size_t)r * 4 + oct) * ip;
    half8 bh, bl;
    float best[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t px = ((size_t)(2 * oy + (k >> 1)) * w + 2 * ox + (k & 1)) * 16;
      const half8 hi = *reinterpret_cast<const half8*>(s + px);
      const half8 lo = *reinterpret_cast<const half8*>(s + 2 * ip + px);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = sp_value(hi, lo, e);
        if (k == 0 || v > best[e]) { best[e] = v; bh[e] = hi[e]; bl[e] = lo[e]; }   // the pair is copied: exact
      }
    }
    unsigned char* d = dst + ((size_t)r * 4 + oct) * op + ((size_t)oy * wo + ox) * 16;
    *reinterpret_cast<half8*>(d) = bh;
    *reinterpret_cast<half8*>(d + 2 * op) = bl;
  }
}

// ATen's upsample_bilinear2d with align_corners=True: src = dst * (in - 1) / (out - 1) in fp32,
// i0 = (int)src, lambda1 = src - i0, value = l0y * (l0x * a + l1x * b) + l1y * (l0x * c + l1x * d)
__global__ void sp_upsample2_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                    int cg_total, int h, int w, long total) {
  const int ho = 2 * h, wo = 2 * w;
  const size_t ip = (size_t)h * w * 16, op = (size_t)ho * wo * 16;
  const float sy = ho > 1 ? (float)(h - 1) / (float)(ho - 1) : 0.f, sx = wo > 1 ? (float)(w - 1) / (float)(wo - 1) : 0.f;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int ox = idx % wo;
    long r = idx / wo;
    const int oy = r % ho; r /= ho;
    const int oct = r % 2; r /= 2;
    const float fy = sy * oy, fx = sx * ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly1 = fy - y0, ly0 = 1.f - ly1, lx1 = fx - x0, lx0 = 1.f - lx1;
    const unsigned char* s = src + ((size_t)r * 4 + oct) * ip;
    auto piece = [&](int y, int x, half8& hi, half8& lo) {
      const size_t px = ((size_t)y * w + x) * 16;
      hi = *reinterpret_cast<const half8*>(s + px);
      lo = *reinterpret_cast<const half8*>(s + 2 * ip + px);
    };
    half8 ah, al, bh, bl, ch, cl, dh, dl, oh, ol;
    piece(y0, x0, ah, al);
    piece(y0, x1, bh, bl);
    piece(y1, x0, ch, cl);
    piece(y1, x1, dh, dl);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = ly0 * (lx0 * sp_value(ah, al, e) + lx1 * sp_value(bh, bl, e)) +
                      ly1 * (lx0 * sp_value(ch, cl, e) + lx1 * sp_value(dh, dl, e));
      _Float16 hi, lo;
      sp_split(v, hi, lo);
      oh[e] = hi;
      ol[e] = lo;
    }
    unsigned char* d = dst + ((size_t)r * 4 + oct) * op + ((size_t)oy * wo + ox) * 16;
    *reinterpret_cast<half8*>(d) = oh;
    *reinterpret_cast<half8*>(d + 2 * op) = ol;
  }
}

// labels [pixels] int32: IGNORE (-100, nn.CrossEntropyLoss's ignore_index) = not counted; any other value
// outside [0, classes) is counted in counts[1] (torch raises there) and otherwise treated as ignored.
__global__ void seg_label_count_kernel(const int32_t* __restrict__ labels, long pixels, int classes,
                                       int32_t* __restrict__ counts) {
  __shared__ int live_s[256], bad_s[256];
  int live = 0, bad = 0;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < pixels; p += (long)gridDim.x * blockDim.x) {
    const int y = labels[p];
    if (y >= 0 && y < classes) ++live;
    else if (y != -100) ++bad;
  }
  live_s[threadIdx.x] = live;
  bad_s[threadIdx.x] = bad;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) { live_s[threadIdx.x] += live_s[threadIdx.x + st]; bad_s[threadIdx.x] += bad_s[threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { atomicAdd(&counts[0], live_s[0]); atomicAdd(&counts[1], bad_s[0]); }   // integer: order-free
}

// logits [pixels][ld] float32 (first `classes` columns), labels [pixels] int32 (outside [0, classes) = ignored).
// loss += sum over live pixels of (logsumexp - z[label]) in double; dlogits = (softmax - onehot) * g with
// g = gscale, or gscale / counts[0] when `counts` is given (mean over the live pixels, as torch).
// CLS > 0: class count known at compile time (logits in registers); CLS == 0: any count, three passes over
// the row (the re-reads hit L1).
template <int CLS>
__global__ void seg_ce_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels, long pixels,
                              int classes, int ld, float gscale, const int32_t* __restrict__ counts,
                              double* __restrict__ loss, float* __restrict__ dlogits) {
  __shared__ double part[256];
  double acc = 0.0;
  if (counts) gscale = counts[0] > 0 ? gscale / (float)counts[0] : 0.f;
  for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < pixels; p += (long)gridDim.x * blockDim.x) {
    const float* z = logits + p * ld;
    const int y = labels[p];
    if constexpr (CLS > 0) {
      float v[CLS], m = -INFINITY;
#pragma unroll
      for (int c = 0; c < CLS; ++c) { v[c] = z[c]; m = fmaxf(m, v[c]); }
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < CLS; ++c) { v[c] = expf(v[c] - m); s += v[c]; }
      const bool live = y >= 0 && y < CLS;
      if (live) acc += (double)(logf(s) + m) - (double)z[y];
      if (dlogits) {
        const float inv = live ? gscale / s : 0.f;
#pragma unroll
        for (int c = 0; c < CLS; ++c) dlogits[p * ld + c] = v[c] * inv - (live && c == y ? gscale : 0.f);
      }
    } else {
      float m = -INFINITY, s = 0.f;
      for (int c = 0; c < classes; ++c) m = fmaxf(m, z[c]);
      for (int c = 0; c < classes; ++c) s += expf(z[c] - m);
      const bool live = y >= 0 && y < classes;
      if (live) acc += (double)(logf(s) + m) - (double)z[y];
      if (dlogits) {
        const float inv = live ? gscale / s : 0.f;
        for (int c = 0; c < classes; ++c)
          dlogits[p * ld + c] = expf(z[c] - m) * inv - (live && c == y ? gscale : 0.f);
      }
    }
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(loss, part[0]);
}

inline int blocks_for(long total) { return (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384); }

}  // namespace

extern "C" int dn_sp_maxpool2(const void* src_sp, int n_images, int h, int w, int channels, void* dst_sp,
                              void* stream) {
  DN_REQUIRE(src_sp && dst_sp && n_images > 0 && channels > 0, "sp_maxpool2: bad arguments");
  DN_REQUIRE(h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0, "sp_maxpool2: %d x %d is not even", h, w);
  const int cg = (channels + 15) / 16;
  const long total = (long)n_images * cg * 2 * (h / 2) * (w / 2);
  hipLaunchKernelGGL(sp_maxpool2_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)src_sp, (unsigned char*)dst_sp, cg, h, w, total);
  return dn::check_launch("sp_maxpool2_kernel");
}

extern "C" int dn_sp_upsample2_bilinear(const void* src_sp, int n_images, int h, int w, int channels,
                                        void* dst_sp, void* stream) {
  DN_REQUIRE(src_sp && dst_sp && n_images > 0 && channels > 0 && h > 0 && w > 0,
             "sp_upsample2_bilinear: bad arguments");
  const int cg = (channels + 15) / 16;
  const long total = (long)n_images * cg * 2 * (2 * h) * (2 * w);
  hipLaunchKernelGGL(sp_upsample2_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)src_sp, (unsigned char*)dst_sp, cg, h, w, total);
  return dn::check_launch("sp_upsample2_kernel");
}

extern "C" int dn_seg_label_count(const int32_t* labels, long pixels, int classes, int32_t* counts,
                                  void* stream) {
  DN_REQUIRE(labels && counts && pixels > 0 && classes > 0, "seg_label_count: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dn::zero_fill(counts, 2 * sizeof(int32_t), s) != hipSuccess)
    return dn::fail(DN_ERR_LAUNCH, "seg_label_count: zero fill failed");
  const int blocks = (int)((pixels + 255) / 256 < 1024 ? (pixels + 255) / 256 : 1024);
  hipLaunchKernelGGL(seg_label_count_kernel, dim3(blocks), dim3(256), 0, s, labels, pixels, classes, counts);
  return dn::check_launch("seg_label_count_kernel");
}

extern "C" int dn_seg_ce_loss(const float* logits, const int32_t* labels, long pixels, int classes, int ld,
                              float grad_scale, const int32_t* counts, double* loss_sum, float* dlogits,
                              void* stream) {
  DN_REQUIRE(logits && labels && loss_sum && pixels > 0, "seg_ce_loss: bad arguments");
  DN_REQUIRE(classes >= 1 && classes <= 1024 && ld >= classes, "seg_ce_loss: %d classes (ld %d) unsupported", classes, ld);
  hipStream_t s = (hipStream_t)stream;
  if (dn::zero_fill(loss_sum, sizeof(double), s) != hipSuccess)
    return dn::fail(DN_ERR_LAUNCH, "seg_ce_loss: zero fill failed");
  const int blocks = (int)((pixels + 255) / 256 < 2048 ? (pixels + 255) / 256 : 2048);
  if (classes == 8)
    hipLaunchKernelGGL(seg_ce_kernel<8>, dim3(blocks), dim3(256), 0, s, logits, labels, pixels, classes, ld, grad_scale,
                       counts, loss_sum, dlogits);
  else
    hipLaunchKernelGGL(seg_ce_kernel<0>, dim3(blocks), dim3(256), 0, s, logits, labels, pixels, classes, ld, grad_scale,
                       counts, loss_sum, dlogits);
  return dn::check_launch("seg_ce_kernel");
}
