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

// ---- training forms on fp32 NHWC maps (the UNet's backward; include/disconet_train.h conventions) ----------
typedef float f32x4s __attribute__((ext_vector_type(4)));

// nn.MaxPool2d(2): y[n][h/2][w/2][c]; one thread per (output pixel, 4 channels)
__global__ void maxpool2_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int h, int w, int c4n, long total) {
  const int ho = h >> 1, wo = w >> 1;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c4 = idx % c4n;
    long r = idx / c4n;
    const int ox = r % wo; r /= wo;
    const int oy = r % ho;
    const long n = r / ho;
    const float* s = x + (((n * h + 2 * oy) * w + 2 * ox) * (long)c4n + c4) * 4;
    f32x4s best = *reinterpret_cast<const f32x4s*>(s);
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const f32x4s v = *reinterpret_cast<const f32x4s*>(s + ((long)(k >> 1) * w + (k & 1)) * c4n * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) best[e] = (v[e] > best[e] || v[e] != v[e]) ? v[e] : best[e];   // ATen: val > max || isnan(val)
    }
    *reinterpret_cast<f32x4s*>(y + idx * 4) = best;
  }
}

// backward of the above: dx[2oy + i][2ox + j] = dy[oy][ox] at the window's arg-max -- the FIRST maximum in scan
// order (0,0) (0,1) (1,0) (1,1), as ATen's forward records it (strict >), so the zero windows of post-ReLU maps
// route their gradient to the top-left pixel exactly like torch -- and 0 at the other three.  Every dx element is
// written exactly once: no zero fill, no atomics, deterministic.  dy: pixel stride ld_dy (a channel slice is fine).
__global__ void maxpool2_nhwc_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int ld_dy,
                                         float* __restrict__ dx, int h, int w, int c4n, long total) {
  const int ho = h >> 1, wo = w >> 1;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c4 = idx % c4n;
    long r = idx / c4n;
    const int ox = r % wo; r /= wo;
    const int oy = r % ho;
    const long n = r / ho;
    const long base = (((n * h + 2 * oy) * w + 2 * ox) * (long)c4n + c4) * 4;
    f32x4s v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4s*>(x + base + ((long)(k >> 1) * w + (k & 1)) * c4n * 4);
    const f32x4s g = *reinterpret_cast<const f32x4s*>(dy + ((n * ho + oy) * wo + ox) * (long)ld_dy + c4 * 4);
    int arg[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float best = v[0][e];
      arg[e] = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k][e] > best || v[k][e] != v[k][e]) { best = v[k][e]; arg[e] = k; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x4s o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = arg[e] == k ? g[e] : 0.f;
      *reinterpret_cast<f32x4s*>(dx + base + ((long)(k >> 1) * w + (k & 1)) * c4n * 4) = o;
    }
  }
}

// nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True): ATen's arithmetic (cf. sp_upsample2_kernel)
struct Lerp { int i0, i1; float l0, l1; };
__device__ inline Lerp lerp_of(int o, int in, float scale) {
  const float f = scale * o;
  Lerp t;
  t.i0 = (int)f;
  t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
  t.l1 = f - t.i0;
  t.l0 = 1.f - t.l1;
  return t;
}

__global__ void upsample2_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int h, int w, int c4n, long total) {
  const int ho = 2 * h, wo = 2 * w;
  const float sy = ho > 1 ? (float)(h - 1) / (float)(ho - 1) : 0.f, sx = wo > 1 ? (float)(w - 1) / (float)(wo - 1) : 0.f;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c4 = idx % c4n;
    long r = idx / c4n;
    const int ox = r % wo; r /= wo;
    const int oy = r % ho;
    const long n = r / ho;
    const Lerp ty = lerp_of(oy, h, sy), tx = lerp_of(ox, w, sx);
    auto at = [&](int yy, int xx) { return *reinterpret_cast<const f32x4s*>(x + (((n * h + yy) * w + xx) * (long)c4n + c4) * 4); };
    const f32x4s a = at(ty.i0, tx.i0), b = at(ty.i0, tx.i1), cc = at(ty.i1, tx.i0), d = at(ty.i1, tx.i1);
    *reinterpret_cast<f32x4s*>(y + idx * 4) = ty.l0 * (tx.l0 * a + tx.l1 * b) + ty.l1 * (tx.l0 * cc + tx.l1 * d);
  }
}

// backward, GATHER form: input pixel (iy, ix) collects dy[oy][ox] * wy(oy -> iy) * wx(ox -> ix) over the output
// rows / columns whose taps reach it -- o in [2 i - 2, 2 i + 2] covers every o with i0(o) or i1(o) == i for a x2
// align_corners map (i0(o) = floor(o (in - 1) / (2 in - 1)) lies in {ceil(o / 2) - 1, floor(o / 2)}) -- each weight
// re-derived with the forward's arithmetic, summed in a fixed order: deterministic, no atomics, no zero fill.
__global__ void upsample2_nhwc_bwd_kernel(const float* __restrict__ dy, int ld_dy, float* __restrict__ dx, int h, int w,
                                          int c4n, long total) {
  const int ho = 2 * h, wo = 2 * w;
  const float sy = ho > 1 ? (float)(h - 1) / (float)(ho - 1) : 0.f, sx = wo > 1 ? (float)(w - 1) / (float)(wo - 1) : 0.f;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c4 = idx % c4n;
    long r = idx / c4n;
    const int ix = r % w; r /= w;
    const int iy = r % h;
    const long n = r / h;
    float wy[5], wx[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int oy = 2 * iy - 2 + k, ox = 2 * ix - 2 + k;
      wy[k] = wx[k] = 0.f;
      if (oy >= 0 && oy < ho) {
        const Lerp t = lerp_of(oy, h, sy);
        wy[k] = (t.i0 == iy ? t.l0 : 0.f) + (t.i1 == iy ? t.l1 : 0.f);
      }
      if (ox >= 0 && ox < wo) {
        const Lerp t = lerp_of(ox, w, sx);
        wx[k] = (t.i0 == ix ? t.l0 : 0.f) + (t.i1 == ix ? t.l1 : 0.f);
      }
    }
    f32x4s acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
      const int oy = 2 * iy - 2 + ky;
      if (wy[ky] == 0.f) continue;
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) {
        const int ox = 2 * ix - 2 + kx;
        if (wx[kx] == 0.f) continue;
        const f32x4s g = *reinterpret_cast<const f32x4s*>(dy + ((n * ho + oy) * wo + ox) * (long)ld_dy + c4 * 4);
        acc += g * (wy[ky] * wx[kx]);
      }
    }
    *reinterpret_cast<f32x4s*>(dx + idx * 4) = acc;
  }
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

// ---- fp32 NHWC training forms (disconet_seg.h) ----
namespace {
int check_nhwc(const char* what, const void* a, const void* b, int n, int h, int w, int c) {
  DN_REQUIRE(a && b && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "%s: bad arguments (c must be a multiple of 4)", what);
  DN_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0, "%s: buffers must be 16-byte aligned", what);
  return DN_OK;
}
}  // namespace

extern "C" int dn_maxpool2_nhwc(const float* x, int n_images, int h, int w, int c, float* y, void* stream) {
  if (int rc = check_nhwc("maxpool2_nhwc", x, y, n_images, h, w, c)) return rc;
  DN_REQUIRE(h % 2 == 0 && w % 2 == 0, "maxpool2_nhwc: %d x %d is not even", h, w);
  const long total = (long)n_images * (h / 2) * (w / 2) * (c / 4);
  hipLaunchKernelGGL(maxpool2_nhwc_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, h, w, c / 4, total);
  return dn::check_launch("maxpool2_nhwc_kernel");
}

extern "C" int dn_maxpool2_nhwc_backward(const float* x, const float* dy, int ld_dy, int n_images, int h, int w, int c,
                                         float* dx, void* stream) {
  if (int rc = check_nhwc("maxpool2_nhwc_backward", x, dx, n_images, h, w, c)) return rc;
  DN_REQUIRE(dy && ld_dy >= c && ld_dy % 4 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && h % 2 == 0 && w % 2 == 0,
             "maxpool2_nhwc_backward: bad gradient view / odd map");
  const long total = (long)n_images * (h / 2) * (w / 2) * (c / 4);
  hipLaunchKernelGGL(maxpool2_nhwc_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, dy, ld_dy, dx, h,
                     w, c / 4, total);
  return dn::check_launch("maxpool2_nhwc_bwd_kernel");
}

extern "C" int dn_upsample2_bilinear_nhwc(const float* x, int n_images, int h, int w, int c, float* y, void* stream) {
  if (int rc = check_nhwc("upsample2_bilinear_nhwc", x, y, n_images, h, w, c)) return rc;
  const long total = (long)n_images * (2 * h) * (2 * w) * (c / 4);
  hipLaunchKernelGGL(upsample2_nhwc_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, h, w, c / 4, total);
  return dn::check_launch("upsample2_nhwc_kernel");
}

extern "C" int dn_upsample2_bilinear_nhwc_backward(const float* dy, int ld_dy, int n_images, int h, int w, int c, float* dx,
                                                   void* stream) {
  if (int rc = check_nhwc("upsample2_bilinear_nhwc_backward", dy, dx, n_images, h, w, c)) return rc;
  DN_REQUIRE(ld_dy >= c && ld_dy % 4 == 0, "upsample2_bilinear_nhwc_backward: bad gradient view");
  const long total = (long)n_images * h * w * (c / 4);
  hipLaunchKernelGGL(upsample2_nhwc_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, dy, ld_dy, dx, h, w,
                     c / 4, total);
  return dn::check_launch("upsample2_nhwc_bwd_kernel");
}
