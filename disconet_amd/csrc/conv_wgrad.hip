// Weight gradient of the conv layers (training step).
//
// Replaces the autograd backward of every nn.Conv2d / Conv3d(1,1,1) on the path
// (upstream:coperception/models/det/backbone/Backbone.py, base/*: conv layers trained
// through loss.backward() in upstream:coperception/utils/CoDetModule.py :: step).
//
//   dW[co][ci][ky][kx] = sum over (image, oy, ox) of
//                        dz[img, oy, ox, co] * x[img, oy*s + ky - pad, ox*s + kx - pad, ci]
//
// A GEMM per tap with K = pixels: D[co][ci] += dz^T . x_shifted on the exact-fp32
// MFMA (v_mfma_f32_32x32x2_f32; K = 2 pixels per instruction).  A workgroup owns a
// 32 x 32 (co, ci) block for all taps and a slice of the pixel tiles: it stages the
// dz tile and the halo patch of x (the same gather as the forward kernel: nearest
// x2 upsample of source 0, channel concat of two sources, stride 2) in LDS, every
// wave takes a quarter of the tile's pixels through all taps (9 independent
// accumulators, no dependent MFMA chain), the four waves meet in an LDS reduction
// and the slice's partial block goes to the workspace.  A second kernel sums the
// slices in a fixed order (deterministic, no float atomics in HBM) into OIHW.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "disconet_train.h"
#include "dn_internal.h"
#include "sp_device.h"

namespace {

struct WgradArgs {
  const float* src0;
  const float* src1;
  const float* dz;
  float* partial;
  int n_images, h_in, w_in, h_out, w_out;
  int c0, c1, up0, c_out;
  int ld0, ld1, ldz;
  int tiles_x, tiles_y, n_tiles;   // pixel tiles per image row / column, total
  int n_cot, n_cit, n_slices;
  int vec0, vec1, vecz;
};

template <int KS, int STRIDE>
struct WgradTile {
  static constexpr int TH = STRIDE == 1 ? 8 : 4;
  static constexpr int TW = 16;
  static constexpr int BM = TH * TW;
  static constexpr int PAD = KS / 2;
  static constexpr int PH = (TH - 1) * STRIDE + KS;
  static constexpr int PW = (TW - 1) * STRIDE + KS;
  static constexpr int TAPS = KS * KS;
  static constexpr int X_VEC = PH * PW * 8;            // float4 slots of the x patch (32 channels)
  static constexpr int X_IT = (X_VEC + 255) / 256;
  static constexpr int D_IT = BM * 8 / 256;
  static constexpr int X_FLOATS = PH * PW * 32;
  static constexpr int D_FLOATS = BM * 32;
  static constexpr int R_FLOATS = TAPS * 1024;         // the block's reduction buffer
  static constexpr int LDS_FLOATS = (X_FLOATS + D_FLOATS) > R_FLOATS ? (X_FLOATS + D_FLOATS) : R_FLOATS;
};

template <int KS, int STRIDE>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradArgs a) {
  using T = WgradTile<KS, STRIDE>;
  constexpr int TH = T::TH, TW = T::TW, PW = T::PW, TAPS = T::TAPS;
  constexpr unsigned OOB = 0xFFFFFFFFu;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;
  float* Ds = smem + T::X_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;

  // work item -> (co block, ci block, slice); slices of one (co, ci) block are adjacent
  int item = blockIdx.x;
  const int slice = item % a.n_slices;
  item /= a.n_slices;
  const int cit = item % a.n_cit;
  const int cot = item / a.n_cit;
  const int co0 = cot * 32, ci0 = cit * 32;
  const bool from1 = ci0 >= a.c0;                       // the ci block lies in the concat source
  const int cs0 = from1 ? ci0 - a.c0 : ci0;             // first channel inside that source
  const int csrc = from1 ? a.c1 : a.c0;                 // channels of that source
  const int ld = from1 ? a.ld1 : a.ld0;
  const bool up = !from1 && a.up0;
  const int hs = up ? a.h_in >> 1 : a.h_in, ws = up ? a.w_in >> 1 : a.w_in;
  const float* src = from1 ? a.src1 : a.src0;
  const bool vecx = from1 ? a.vec1 : a.vec0;
  const size_t img_x = (size_t)hs * ws * ld, img_z = (size_t)a.h_out * a.w_out * a.ldz;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  f32x4 rx[T::X_IT], rd[T::D_IT];

  auto ld128 = [](auto rsrc, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
  };
  // channel tails / unaligned rows: dword loads, each with its own bound
  auto ld32x4 = [](auto rsrc, unsigned voff, int nvalid) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                           rsrc, (voff == 0xFFFFFFFFu || e >= nvalid) ? 0xFFFFFFFFu : voff + 4 * e, 0, 0));
    return v;
  };

  auto load_tile = [&](int tile) {
    int sp = tile;
    const int ox0 = (sp % a.tiles_x) * TW;
    sp /= a.tiles_x;
    const int oy0 = (sp % a.tiles_y) * TH;
    const int img = sp / a.tiles_y;
    const auto rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src + img * img_x), 0,
                                                       (int)(img_x * 4), 0x00020000);
    const auto rsz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz + img * img_z), 0,
                                                       (int)(img_z * 4), 0x00020000);
    const int iy0 = oy0 * STRIDE - T::PAD, ix0 = ox0 * STRIDE - T::PAD;
#pragma unroll
    for (int it = 0; it < T::X_IT; ++it) {
      const int idx = tid + it * 256;
      const int p = idx >> 3, q = idx & 7;
      const int iy = iy0 + p / PW, ix = ix0 + p % PW;
      const int c = cs0 + 4 * q;
      const bool ok = idx < T::X_VEC && iy >= 0 && iy < a.h_in && ix >= 0 && ix < a.w_in && c < csrc;
      const int sy = up ? iy >> 1 : iy, sx = up ? ix >> 1 : ix;
      const unsigned off = ok ? (unsigned)(((sy * ws + sx) * ld + c) * 4) : OOB;
      rx[it] = vecx ? ld128(rsx, off) : ld32x4(rsx, off, csrc - c);
    }
#pragma unroll
    for (int it = 0; it < T::D_IT; ++it) {
      const int idx = tid + it * 256;
      const int m = idx >> 3, q = idx & 7;
      const int oy = oy0 + m / TW, ox = ox0 + m % TW;
      const int c = co0 + 4 * q;
      const bool ok = oy < a.h_out && ox < a.w_out && c < a.c_out;
      const unsigned off = ok ? (unsigned)(((oy * a.w_out + ox) * a.ldz + c) * 4) : OOB;
      rd[it] = a.vecz ? ld128(rsz, off) : ld32x4(rsz, off, a.c_out - c);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int it = 0; it < T::X_IT; ++it) {
      const int idx = tid + it * 256;
      if (idx < T::X_VEC) *reinterpret_cast<f32x4*>(&Xs[idx * 4]) = rx[it];
    }
#pragma unroll
    for (int it = 0; it < T::D_IT; ++it)
      *reinterpret_cast<f32x4*>(&Ds[(tid + it * 256) * 4]) = rd[it];
  };

  int tile = slice;
  if (tile < a.n_tiles) {
    load_tile(tile);
    store_tile();
  }
  __syncthreads();
  for (; tile < a.n_tiles; tile += a.n_slices) {
    const bool more = tile + a.n_slices < a.n_tiles;
    if (more) load_tile(tile + a.n_slices);
    // this wave's pixels: rows wave*TH/4 .. of the tile, pairs (x, x + 1) are one K = 2 step
#pragma unroll
    for (int rr = 0; rr < TH / 4; ++rr) {
      const int row = wave * (TH / 4) + rr;
#pragma unroll 2
      for (int kp = 0; kp < TW / 2; ++kp) {
        const int col = 2 * kp + lh;
        const float av = Ds[(row * TW + col) * 32 + li];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          const int ty = t / KS, tx = t % KS;
          const float bv = Xs[((row * STRIDE + ty) * PW + col * STRIDE + tx) * 32 + li];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) store_tile();
    __syncthreads();
  }

  // four waves -> one block: LDS reduction, then the slice's partial goes out row-major
  // (the waves add in wave order, one after the other: a fixed summation order -- round 2 used LDS float atomics,
  // whose arrival order made the weight gradients differ from run to run in the last bits)
  float* R = smem;
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = lh * 4 + (r & 3) + 8 * (r >> 2);
          float* slot = &R[t * 1024 + i * 32 + li];
          *slot = (wv == 0 ? 0.f : *slot) + acc[t][r];
        }
    }
    __syncthreads();
  }
  float* out = a.partial + ((size_t)(slice * a.n_cot + cot) * a.n_cit + cit) * T::R_FLOATS;
  for (int i = tid * 4; i < T::R_FLOATS; i += 1024)
    *reinterpret_cast<f32x4*>(out + i) = *reinterpret_cast<const f32x4*>(&R[i]);
}

// 3x3 stride-1 layers with >= 64 channels on both sides: a 64 x 64 (co, ci) block per workgroup.
// Wave (wm, wn) owns one 32 x 32 quadrant over ALL pixels of the (4 x 16) tile, so the dz tile and
// the x patch are staged once for four times the MFMA work of the 32 x 32 form (half the global
// re-reads, no cross-wave reduction).  LDS rows are 64 floats = all 64 banks: the second half-wave
// (the other pixel of the K = 2 pair) would hit the first one's banks, so odd pixels store their two
// 32-channel halves swapped (c ^ 32) -- conflict-free without padding.
__global__ __launch_bounds__(256, 2) void conv_wgrad64_kernel(const WgradArgs a) {
  constexpr int TH = 4, TW = 16, BM = TH * TW, PH = TH + 2, PW = TW + 2, TAPS = 9;
  constexpr int X_VEC = PH * PW * 16, X_IT = (X_VEC + 255) / 256, D_IT = BM * 16 / 256;
  constexpr unsigned OOB = 0xFFFFFFFFu;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;
  float* Ds = smem + PH * PW * 64;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  int item = blockIdx.x;
  const int slice = item % a.n_slices;
  item /= a.n_slices;
  const int cit = item % a.n_cit;
  const int cot = item / a.n_cit;
  const int co0 = cot * 64, ci0 = cit * 64;
  const bool from1 = ci0 >= a.c0;
  const int cs0 = from1 ? ci0 - a.c0 : ci0;
  const int csrc = from1 ? a.c1 : a.c0;
  const int ld = from1 ? a.ld1 : a.ld0;
  const bool up = !from1 && a.up0;
  const int hs = up ? a.h_in >> 1 : a.h_in, ws = up ? a.w_in >> 1 : a.w_in;
  const float* src = from1 ? a.src1 : a.src0;
  const size_t img_x = (size_t)hs * ws * ld, img_z = (size_t)a.h_out * a.w_out * a.ldz;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  f32x4 rx[X_IT], rd[D_IT];

  auto ld128 = [](auto rsrc, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
  };
  auto load_tile = [&](int tile) {
    int sp = tile;
    const int ox0 = (sp % a.tiles_x) * TW;
    sp /= a.tiles_x;
    const int oy0 = (sp % a.tiles_y) * TH;
    const int img = sp / a.tiles_y;
    const auto rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src + img * img_x), 0,
                                                       (int)(img_x * 4), 0x00020000);
    const auto rsz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz + img * img_z), 0,
                                                       (int)(img_z * 4), 0x00020000);
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int idx = tid + it * 256;
      const int p = idx >> 4, q = idx & 15;
      const int iy = oy0 - 1 + p / PW, ix = ox0 - 1 + p % PW;
      const int c = cs0 + 4 * q;
      const bool ok = idx < X_VEC && iy >= 0 && iy < a.h_in && ix >= 0 && ix < a.w_in && c < csrc;
      const int sy = up ? iy >> 1 : iy, sx = up ? ix >> 1 : ix;
      rx[it] = ld128(rsx, ok ? (unsigned)(((sy * ws + sx) * ld + c) * 4) : OOB);
    }
#pragma unroll
    for (int it = 0; it < D_IT; ++it) {
      const int idx = tid + it * 256;
      const int m = idx >> 4, q = idx & 15;
      const int oy = oy0 + m / TW, ox = ox0 + m % TW;
      const int c = co0 + 4 * q;
      const bool ok = oy < a.h_out && ox < a.w_out && c < a.c_out;
      rd[it] = ld128(rsz, ok ? (unsigned)(((oy * a.w_out + ox) * a.ldz + c) * 4) : OOB);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int idx = tid + it * 256;
      const int p = idx >> 4, q = idx & 15;
      if (idx < X_VEC) *reinterpret_cast<f32x4*>(&Xs[p * 64 + ((4 * q) ^ (((p % PW) & 1) << 5))]) = rx[it];
    }
#pragma unroll
    for (int it = 0; it < D_IT; ++it) {
      const int idx = tid + it * 256;
      const int m = idx >> 4, q = idx & 15;
      *reinterpret_cast<f32x4*>(&Ds[m * 64 + ((4 * q) ^ ((m & 1) << 5))]) = rd[it];
    }
  };

  int tile = slice;
  if (tile < a.n_tiles) {
    load_tile(tile);
    store_tile();
  }
  __syncthreads();
  const int a_col = (wm * 32 + li) ^ (lh << 5);            // dz column of this lane (pixel parity = lh)
  for (; tile < a.n_tiles; tile += a.n_slices) {
    const bool more = tile + a.n_slices < a.n_tiles;
    if (more) load_tile(tile + a.n_slices);
#pragma unroll
    for (int row = 0; row < TH; ++row) {
#pragma unroll 1   // 9 independent MFMAs per step already fill the pipe; more unrolling spills
      for (int kp = 0; kp < TW / 2; ++kp) {
        const int col = 2 * kp + lh;
        const float av = Ds[(row * TW + col) * 64 + a_col];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          const int ty = t / 3, tx = t % 3;
          const int xx = col + tx;
          const float bv = Xs[((row + ty) * PW + xx) * 64 + ((wn * 32 + li) ^ ((xx & 1) << 5))];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (more) store_tile();
    __syncthreads();
  }

  float* out = a.partial + ((size_t)(slice * a.n_cot + cot) * a.n_cit + cit) * (TAPS * 4096);
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = lh * 4 + (r & 3) + 8 * (r >> 2);
      out[t * 4096 + (wm * 32 + i) * 64 + wn * 32 + li] = acc[t][r];
    }
}

// partial [slice][cot][cit][tap][ct co][ct ci] -> dW [c_out][c_in][taps] (OIHW), slices summed in a FIXED order: a workgroup owns
// 64 consecutive elements; wave g adds slices g, g + 4, g + 8, ... (two alternating chains, each wave-load 256 contiguous bytes),
// then the four waves' sums meet in LDS as (w0 + w1) + (w2 + w3).  (Round 5: one thread per element walked all -- up to 512 --
// slices alone; the launches behind the 256 x 256 layers were latency-bound at 20-40 us each.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                           int n_slices, int n_cot, int n_cit, int taps, int c_out, int c_in,
                                                           int cin_total, int accumulate, int ct, float scale) {
  __shared__ float red[4][64];
  const long per_slice = (long)n_cot * n_cit * taps * ct * ct;        // a multiple of 64 (ct * ct is)
  const int i = threadIdx.x & 63, g = threadIdx.x >> 6;
  for (long base = blockIdx.x * 64L; base < per_slice; base += gridDim.x * 64L) {
    const long idx = base + i;
    float s0 = 0.f, s1 = 0.f;
    int sl = g;
    for (; sl + 4 < n_slices; sl += 8) {
      s0 += partial[sl * per_slice + idx];
      s1 += partial[(sl + 4) * per_slice + idx];
    }
    if (sl < n_slices) s0 += partial[sl * per_slice + idx];
    red[g][i] = s0 + s1;
    __syncthreads();
    if (g == 0) {
      const float s = ((red[0][i] + red[1][i]) + (red[2][i] + red[3][i])) * scale;   // scale: 1 (fp32 kernels) or 1 / (dz_lift * x_lift), a power of two: exact
      const int j = (int)(idx % ct), ii = (int)((idx / ct) % ct);
      long r = idx / (ct * ct);
      const int t = (int)(r % taps);
      r /= taps;
      const int cit = (int)(r % n_cit), cot = (int)(r / n_cit);
      const int co = cot * ct + ii, ci = cit * ct + j;
      if (co < c_out && ci < c_in) {
        float* dst = dw + ((size_t)co * cin_total + ci) * taps + t;
        *dst = accumulate ? *dst + s : s;
      }
    }
    __syncthreads();
  }
}

#include "wgrad_sp.inl"

int out_dim(int in, int k, int stride) { return (in + 2 * (k / 2) - k) / stride + 1; }

int validate(const dn_conv_desc* d) {
  DN_REQUIRE(d != nullptr, "wgrad: null descriptor");
  DN_REQUIRE(d->n_images > 0 && d->h_in > 0 && d->w_in > 0 && d->c0 > 0 && d->c1 >= 0 && d->c_out > 0,
             "wgrad: non-positive dimension");
  DN_REQUIRE((d->ksize == 3 && (d->stride == 1 || d->stride == 2)) || (d->ksize == 1 && d->stride == 1),
             "wgrad: ksize %d stride %d unsupported (3x3 s1/s2, 1x1 s1)", d->ksize, d->stride);
  DN_REQUIRE(d->up0 == 0 || (d->up0 == 1 && d->h_in % 2 == 0 && d->w_in % 2 == 0),
             "wgrad: up0 needs even input dims");
  DN_REQUIRE(d->c1 == 0 || d->c0 % 32 == 0, "wgrad: concat needs c0 %% 32 == 0 (got %d)", d->c0);
  DN_REQUIRE(d->c0 + d->c1 <= 4096 && d->c_out <= 4096, "wgrad: channel count out of range");
  DN_REQUIRE(d->ld0 >= d->c0 && d->ld1 >= d->c1 && d->ldo >= d->c_out, "wgrad: row stride < channels");
  return DN_OK;
}

struct Plan {
  int h_out, w_out, tiles_x, tiles_y, n_tiles, n_cot, n_cit, n_slices, taps, ct;
};

Plan make_plan(const dn_conv_desc& d) {
  Plan p;
  // 64 x 64 channel blocks where both sides have them (and every source is 16-byte loadable)
  p.ct = (d.ksize == 3 && d.stride == 1 && d.c_out >= 64 && d.c_out % 4 == 0 && d.c0 % 64 == 0 &&
          d.c1 % 64 == 0 && d.ld0 % 4 == 0 && d.ld1 % 4 == 0 && d.ldo % 4 == 0)
             ? 64 : 32;
  const int th = (d.stride == 1 && p.ct == 32) ? 8 : 4, tw = 16;
  p.h_out = out_dim(d.h_in, d.ksize, d.stride);
  p.w_out = out_dim(d.w_in, d.ksize, d.stride);
  p.tiles_x = (p.w_out + tw - 1) / tw;
  p.tiles_y = (p.h_out + th - 1) / th;
  p.n_tiles = d.n_images * p.tiles_x * p.tiles_y;
  p.n_cot = (d.c_out + p.ct - 1) / p.ct;
  // a concat layer's ci blocks never straddle the sources (c0 % ct == 0)
  p.n_cit = (d.c0 + p.ct - 1) / p.ct + (d.c1 + p.ct - 1) / p.ct;
  p.taps = d.ksize * d.ksize;
  // one resident generation (2 workgroups per CU), every slice at least 4 pixel tiles long;
  // fewer, longer slices also mean fewer partial blocks for the reduction to read back
  int s = 512 / (p.n_cot * p.n_cit);
  if (s > p.n_tiles / 4) s = p.n_tiles / 4;
  if (s < 1) s = 1;
  p.n_slices = s;
  return p;
}

template <int KS, int STRIDE>
int launch(const WgradArgs& a, const Plan& p, hipStream_t stream) {
  using T = WgradTile<KS, STRIDE>;
  auto kern = conv_wgrad_kernel<KS, STRIDE>;
  static dn::PerDeviceFlag ready_flag;
  bool& ready = ready_flag.here();
  if (!ready) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_FLOATS * 4);
    if (e != hipSuccess)
      return dn::fail(DN_ERR_LAUNCH, "wgrad: cannot reserve %d B of LDS: %s", T::LDS_FLOATS * 4,
                      hipGetErrorString(e));
    ready = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.n_cot * p.n_cit * p.n_slices), dim3(256), T::LDS_FLOATS * 4, stream, a);
  return dn::check_launch("conv_wgrad_kernel");
}

}  // namespace

extern "C" size_t dn_conv_wgrad_workspace(const dn_conv_desc* d) {
  if (validate(d) != DN_OK) return 0;
  const Plan p = make_plan(*d);
  return (size_t)p.n_slices * p.n_cot * p.n_cit * p.taps * p.ct * p.ct * sizeof(float);
}

extern "C" int dn_conv_wgrad(const dn_conv_desc* d, const float* src0, const float* src1, const float* dz,
                             void* workspace, float* dw_oihw, int dw_cin_total, int accumulate,
                             void* stream) {
  if (int rc = validate(d)) return rc;
  DN_REQUIRE(src0 && dz && workspace && dw_oihw, "wgrad: null pointer");
  DN_REQUIRE(dw_cin_total == 0 || dw_cin_total >= d->c0 + d->c1, "wgrad: dw_cin_total %d < c_in",
             dw_cin_total);
  DN_REQUIRE(d->c1 == 0 || src1, "wgrad: c1 > 0 needs src1");
  const Plan p = make_plan(*d);
  WgradArgs a;
  a.src0 = src0; a.src1 = src1; a.dz = dz; a.partial = static_cast<float*>(workspace);
  a.n_images = d->n_images; a.h_in = d->h_in; a.w_in = d->w_in; a.h_out = p.h_out; a.w_out = p.w_out;
  a.c0 = d->c0; a.c1 = d->c1; a.up0 = d->up0; a.c_out = d->c_out;
  a.ld0 = d->ld0; a.ld1 = d->ld1; a.ldz = d->ldo;
  a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.n_tiles = p.n_tiles;
  a.n_cot = p.n_cot; a.n_cit = p.n_cit; a.n_slices = p.n_slices;
  auto aligned = [](const void* ptr, int c, int ldv) {
    return c % 4 == 0 && ldv % 4 == 0 && (reinterpret_cast<uintptr_t>(ptr) & 15) == 0;
  };
  a.vec0 = aligned(src0, d->c0, d->ld0);
  a.vec1 = d->c1 ? aligned(src1, d->c1, d->ld1) : 1;
  a.vecz = aligned(dz, d->c_out, d->ldo);
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (p.ct == 64) {
    // the vector path also needs 16-byte aligned bases
    DN_REQUIRE(a.vec0 && a.vec1 && a.vecz, "wgrad: 64-channel blocks need 16-byte aligned sources");
    constexpr int lds = (6 * 18 * 64 + 64 * 64) * 4;
    static dn::PerDeviceFlag ready_flag;
    bool& ready = ready_flag.here();
    if (!ready) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad64_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess)
        return dn::fail(DN_ERR_LAUNCH, "wgrad: cannot reserve %d B of LDS: %s", lds, hipGetErrorString(e));
      ready = true;
    }
    hipLaunchKernelGGL(conv_wgrad64_kernel, dim3(p.n_cot * p.n_cit * p.n_slices), dim3(256), lds, s, a);
    rc = dn::check_launch("conv_wgrad64_kernel");
  } else if (d->ksize == 3 && d->stride == 1) rc = launch<3, 1>(a, p, s);
  else if (d->ksize == 3) rc = launch<3, 2>(a, p, s);
  else rc = launch<1, 1>(a, p, s);
  if (rc) return rc;
  const long per_slice = (long)p.n_cot * p.n_cit * p.taps * p.ct * p.ct;
  const int blocks = (int)(per_slice / 64 < 8192 ? per_slice / 64 : 8192);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, a.partial, dw_oihw, p.n_slices,
                     p.n_cot, p.n_cit, p.taps, d->c_out, d->c0 + d->c1,
                     dw_cin_total ? dw_cin_total : d->c0 + d->c1, accumulate, p.ct, 1.0f);
  return dn::check_launch("wgrad_reduce_kernel");
}

// ---- split-f16 form (wgrad_sp.inl) ------------------------------------------------------------------------
namespace {
// channels per side of a workgroup's block: 64, 32, or 0 = the layer has no split-f16 kernel
int sp_block(const dn_conv_desc& d) {
  if (d.ksize != 3 || (d.stride != 1 && d.stride != 2) || d.c_out % 4 != 0 || d.ldo % 4 != 0) return 0;
  const bool vec = d.c0 % 4 == 0 && d.ld0 % 4 == 0 && d.c1 % 4 == 0 && (d.c1 == 0 || d.ld1 % 4 == 0);
  if (d.stride == 2) {      // one 16-byte loadable source, no upsample (the encoder's four down-sampling layers)
    if (!vec || d.c1 != 0 || d.up0 != 0) return 0;
    return d.c_out >= 64 && d.c0 % 64 == 0 ? 64 : d.c_out >= 32 && d.c0 % 32 == 0 ? 32 : 0;
  }
  if (vec && d.c_out >= 64 && d.c0 % 64 == 0 && d.c1 % 64 == 0) return 64;
  // 32 x 32 blocks; a single source may end in a partial block and need not be 16-byte loadable (the 13-channel voxel grid)
  if (d.c_out >= 32 && (d.c1 == 0 || (vec && d.c0 % 32 == 0))) return 32;
  return 0;
}
Plan make_plan_sp(const dn_conv_desc& d, int cb) {
  Plan p;
  p.ct = cb;
  p.h_out = out_dim(d.h_in, 3, d.stride);
  p.w_out = out_dim(d.w_in, 3, d.stride);
  const int th = d.stride == 2 ? (cb == 64 ? WspShape2<64>::TH : WspShape2<32>::TH) : (cb == 64 ? WspShape<64>::TH : WspShape<32>::TH);
  p.tiles_x = (p.w_out + 15) / 16;
  p.tiles_y = (p.h_out + th - 1) / th;
  p.n_tiles = d.n_images * p.tiles_x * p.tiles_y;
  p.n_cot = (d.c_out + cb - 1) / cb;
  p.n_cit = (d.c0 + cb - 1) / cb + (d.c1 + cb - 1) / cb;
  p.taps = 9;
  // one resident generation (2 workgroups per CU).  The slice is the fast index of a work item and workgroups go round the
  // 8 XCDs by index: with a multiple of 8 slices every (co, ci) block of one slice -- the workgroups that read the same pixel
  // tiles at about the same time -- sits on ONE XCD, behind one L2.
  int s = 512 / (p.n_cot * p.n_cit);
  if (s > p.n_tiles / 4) s = p.n_tiles / 4;
  if (s >= 8) s &= ~7;
  if (s < 1) s = 1;
  p.n_slices = s;
  return p;
}
template <int CB, int STRIDE, bool ZSP = false>
int launch_sp(const WgradSpArgs& a, const Plan& p, hipStream_t stream) {
  constexpr int lds = (STRIDE == 2 ? WspShape2<CB>::LDS_DWORDS : WspShape<CB>::LDS_DWORDS) * 4;
  auto kern = STRIDE == 2 ? conv_wgrad_sp_s2_kernel<CB, ZSP> : conv_wgrad_sp_kernel<CB, ZSP>;
  static dn::PerDeviceFlag ready_flag;
  bool& ready = ready_flag.here();
  if (!ready) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return dn::fail(DN_ERR_LAUNCH, "wgrad_sp: cannot reserve %d B of LDS: %s", lds, hipGetErrorString(e));
    ready = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.n_cot * p.n_cit * p.n_slices), dim3(256), lds, stream, a);
  return dn::check_launch("conv_wgrad_sp_kernel");
}
}  // namespace

extern "C" int dn_conv_wgrad_sp_supported(const dn_conv_desc* d) {
  return d != nullptr && validate(d) == DN_OK ? sp_block(*d) : 0;
}

extern "C" size_t dn_conv_wgrad_sp_workspace(const dn_conv_desc* d) {
  const int cb = dn_conv_wgrad_sp_supported(d);
  if (!cb) return 0;
  const Plan p = make_plan_sp(*d, cb);
  return (size_t)p.n_slices * p.n_cot * p.n_cit * 9 * cb * cb * sizeof(float);
}

namespace {
int conv_wgrad_sp_impl(const dn_conv_desc* d, const float* src0, const float* src1, const float* dz, const void* dz_sp, void* workspace,
                       float* dw_oihw, int dw_cin_total, int accumulate, float dz_lift, float x_lift, void* stream);
}
extern "C" int dn_conv_wgrad_sp(const dn_conv_desc* d, const float* src0, const float* src1, const float* dz, void* workspace,
                                float* dw_oihw, int dw_cin_total, int accumulate, float dz_lift, float x_lift, void* stream) {
  return conv_wgrad_sp_impl(d, src0, src1, dz, nullptr, workspace, dw_oihw, dw_cin_total, accumulate, dz_lift, x_lift, stream);
}
extern "C" int dn_conv_wgrad_sp_z(const dn_conv_desc* d, const float* src0, const float* src1, const void* dz_sp, void* workspace,
                                  float* dw_oihw, int dw_cin_total, int accumulate, float dz_lift, float x_lift, void* stream) {
  DN_REQUIRE(dz_sp, "wgrad_sp_z: null pointer");
  return conv_wgrad_sp_impl(d, src0, src1, nullptr, dz_sp, workspace, dw_oihw, dw_cin_total, accumulate, dz_lift, x_lift, stream);
}
namespace {
int conv_wgrad_sp_impl(const dn_conv_desc* d, const float* src0, const float* src1, const float* dz, const void* dz_sp, void* workspace,
                       float* dw_oihw, int dw_cin_total, int accumulate, float dz_lift, float x_lift, void* stream) {
  if (int rc = validate(d)) return rc;
  const int cb = sp_block(*d);
  DN_REQUIRE(cb != 0, "wgrad_sp: 3x3 layers with c_out >= 32 only; two sources: 32 k channels each, stride 1 (dn_conv_wgrad_sp_supported)");
  DN_REQUIRE(src0 && (dz || dz_sp) && workspace && dw_oihw, "wgrad_sp: null pointer");
  DN_REQUIRE(!dz_sp || (d->c_out % 16 == 0 && (reinterpret_cast<uintptr_t>(dz_sp) & 15) == 0),
             "wgrad_sp_z: the SP copy of dz needs c_out %% 16 == 0 (got %d) and a 16-byte aligned tensor", d->c_out);
  DN_REQUIRE(dw_cin_total == 0 || dw_cin_total >= d->c0 + d->c1, "wgrad_sp: dw_cin_total %d < c_in", dw_cin_total);
  DN_REQUIRE(d->c1 == 0 || src1, "wgrad_sp: c1 > 0 needs src1");
  auto pow2 = [](float v) { int e; return v > 0.f && std::isfinite(v) && std::frexp(v, &e) == 0.5f; };
  DN_REQUIRE(pow2(dz_lift) && pow2(x_lift), "wgrad_sp: the lifts must be powers of two (got %g, %g)", dz_lift, x_lift);
  auto al16 = [](const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
  const bool vecx = d->c0 % 4 == 0 && d->ld0 % 4 == 0 && al16(src0) && (d->c1 == 0 || (d->c1 % 4 == 0 && d->ld1 % 4 == 0 && al16(src1)));
  DN_REQUIRE((dz_sp || al16(dz)) && (vecx || (cb == 32 && d->c1 == 0 && d->stride == 1)), "wgrad_sp: dz (and, but for a single-source stride-1 "
             "32-channel-block layer, the sources) must be 16-byte aligned with rows of 4 k floats");
  const Plan p = make_plan_sp(*d, cb);
  WgradSpArgs a;
  a.src0 = src0; a.src1 = src1; a.dz = dz; a.partial = static_cast<float*>(workspace);
  a.n_images = d->n_images; a.h_in = d->h_in; a.w_in = d->w_in;
  a.c0 = d->c0; a.c1 = d->c1; a.up0 = d->up0; a.c_out = d->c_out;
  a.ld0 = d->ld0; a.ld1 = d->ld1; a.ldz = d->ldo;
  a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y; a.n_tiles = p.n_tiles;
  a.n_cot = p.n_cot; a.n_cit = p.n_cit; a.n_slices = p.n_slices;
  a.dz_lift = dz_lift; a.x_lift = x_lift; a.vecx = vecx ? 1 : 0;
  a.dz_sp = static_cast<const unsigned char*>(dz_sp);
  hipStream_t s = (hipStream_t)stream;
  if (int rc = dz_sp ? (d->stride == 2 ? (cb == 64 ? launch_sp<64, 2, true>(a, p, s) : launch_sp<32, 2, true>(a, p, s))
                                       : (cb == 64 ? launch_sp<64, 1, true>(a, p, s) : launch_sp<32, 1, true>(a, p, s)))
                     : (d->stride == 2 ? (cb == 64 ? launch_sp<64, 2>(a, p, s) : launch_sp<32, 2>(a, p, s))
                                       : (cb == 64 ? launch_sp<64, 1>(a, p, s) : launch_sp<32, 1>(a, p, s))))
    return rc;
  const long per_slice = (long)p.n_cot * p.n_cit * 9 * cb * cb;
  const int blocks = (int)(per_slice / 64 < 8192 ? per_slice / 64 : 8192);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, a.partial, dw_oihw, p.n_slices, p.n_cot, p.n_cit, 9,
                     d->c_out, d->c0 + d->c1, dw_cin_total ? dw_cin_total : d->c0 + d->c1, accumulate, cb,
                     1.0f / (dz_lift * x_lift));
  return dn::check_launch("wgrad_reduce_kernel");
}
}  // namespace

namespace dn { void range_collect_conv_wgrad(unsigned* dst, bool reset, hipStream_t s) { sp_range_collect_here(dst, reset, s); } }

namespace {
// wt[ci][co][kk - 1 - t] = w[co][ci_first + ci][t]
__global__ void dgrad_weights_kernel(const float* __restrict__ w, int c_out, int cin_total, int ci_first,
                                     int c_in, int taps, float* __restrict__ wt) {
  const long total = (long)c_in * c_out * taps;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int t = (int)(idx % taps);
    const long r = idx / taps;
    const int co = (int)(r % c_out), ci = (int)(r / c_out);
    wt[idx] = w[((size_t)co * cin_total + ci_first + ci) * taps + (taps - 1 - t)];
  }
}
}  // namespace

namespace {
// Weights of ONE parity class of the data gradient of a STRIDE-2 3x3 layer (the "parity-phase" form).  Input pixel
// i = 2 m + p of the forward layer receives  dx[i] = sum over taps t with (i - t + 1) even of  W[t]^T dz[(i - t + 1) / 2]:
//   p = 0:  t = 1             -> dz[m]
//   p = 1:  t = 0 -> dz[m + 1],  t = 2 -> dz[m]
// per axis.  As a 3x3 stride-1 conv over dz (taps v reading dz[m + v - 1]) that is v[1] = W[1] (p = 0) and v[1] = W[2],
// v[2] = W[0] (p = 1); every other tap is zero and is masked out of the launch (dn_conv2d_taps).  One class touches 1, 2,
// 2 or 4 of the 9 taps: a quarter of the MFMAs of the zero-stuffed form (which runs all 9 taps on every input pixel).
// v[ci][co][vy][vx] <- w[co][ci_first + ci][ty][tx], zero elsewhere
__global__ void dgrad_class_weights_kernel(const float* __restrict__ w, int c_out, int cin_total, int ci_first, int c_in,
                                           int py, int px, float* __restrict__ v) {
  const long total = (long)c_in * c_out * 9;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int t = (int)(idx % 9), vy = t / 3, vx = t % 3;
    const long r = idx / 9;
    const int co = (int)(r % c_out), ci = (int)(r / c_out);
    auto src_tap = [](int p, int vv) { return p == 0 ? (vv == 1 ? 1 : -1) : (vv == 1 ? 2 : vv == 2 ? 0 : -1); };
    const int ty = src_tap(py, vy), tx = src_tap(px, vx);
    v[idx] = (ty < 0 || tx < 0) ? 0.f : w[((size_t)co * cin_total + ci_first + ci) * 9 + ty * 3 + tx];
  }
}
}  // namespace

extern "C" int dn_conv_dgrad_class_weights(const float* w_oihw, int c_out, int cin_total, int ci_first, int c_in, int py,
                                           int px, float* v_oihw, int* tap_mask, void* stream) {
  DN_REQUIRE(w_oihw && v_oihw && tap_mask, "dgrad class weights: null pointer");
  DN_REQUIRE(c_out > 0 && c_in > 0 && ci_first >= 0 && ci_first + c_in <= cin_total && (py | px) >= 0 && py <= 1 && px <= 1,
             "dgrad class weights: bad shape / parity");
  const long total = (long)c_in * c_out * 9;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(dgrad_class_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, c_out, cin_total,
                     ci_first, c_in, py, px, v_oihw);
  const int rows = py == 0 ? 0b010 : 0b110, cols = px == 0 ? 0b010 : 0b110;     // bit v set: tap row / column v is used
  int m = 0;
  for (int vy = 0; vy < 3; ++vy)
    for (int vx = 0; vx < 3; ++vx)
      if (((rows >> vy) & 1) && ((cols >> vx) & 1)) m |= 1 << (vy * 3 + vx);
  *tap_mask = m;
  return dn::check_launch("dgrad_class_weights_kernel");
}

extern "C" int dn_conv_dgrad_weights(const float* w_oihw, int c_out, int cin_total, int ci_first,
                                     int c_in, int ksize, float* wt_oihw, void* stream) {
  DN_REQUIRE(w_oihw && wt_oihw, "dgrad weights: null pointer");
  DN_REQUIRE(c_out > 0 && c_in > 0 && ci_first >= 0 && ci_first + c_in <= cin_total &&
                 (ksize == 1 || ksize == 3),
             "dgrad weights: bad shape");
  const long total = (long)c_in * c_out * ksize * ksize;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(dgrad_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, c_out,
                     cin_total, ci_first, c_in, ksize * ksize, wt_oihw);
  return dn::check_launch("dgrad_weights_kernel");
}
