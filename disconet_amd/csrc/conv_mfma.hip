// Implicit-GEMM convolution for gfx950 on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32), NHWC activations, fused affine (+ReLU) epilogue,
// optional on-the-fly "nearest x2 upsample of src0, concat src1" input.
//
// GEMM view: M = output pixels (a TH x TW spatial tile per workgroup),
// N = output channels (BN per workgroup), K = taps x input channels.
// Per input-channel chunk (KC channels) the workgroup stages ONE halo patch
// [(TH-1)*S+KS][(TW-1)*S+KS][KC] and the matching weights [taps][BN][KC] in
// LDS; the KS*KS taps then read the same patch at shifted pixel offsets, so
// the input is fetched once per chunk instead of once per tap (no im2col
// materialisation, no 9x re-read).
//
// MFMA operand mapping (cdna_hip_programming.md §3): for 32x32x2 lane l holds
// A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].  Both LDS images keep K
// contiguous per row, so one ds_read_b128 per operand feeds four MFMAs: lane
// (i, h) reads k = 8s + 4h + {0..3}; the K order inside the chunk is a
// permutation shared by A and B, which only changes the fp32 summation order.
// Row stride KC+4 floats keeps every 16-lane ds_read_b128 group on 16
// distinct 16-byte slots (conflict-free for stride 1).
//
// Replaces the conv2d/conv3d(1x1x1)+batch_norm+relu(+interpolate+cat) chains of
// upstream:coperception/models/det/backbone/Backbone.py.
#include "dn_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct ConvArgs {
  const float* src0;
  const float* src1;
  const float* wpk;
  const float* scale;
  const float* shift;
  float* out;
  int n_images, h_in, w_in, h_out, w_out;
  int c0, c1, up0, c_out, relu;
  int ld0, ld1, ldo;
  int nchunks, tiles_x, tiles_y;
  int vec0, vec1;
};

template <int KS, int STRIDE, int TH, int TW, int BN, int KC, int WAVES_M, int WAVES_N,
          int WTM, int WTN>
struct ConvTile {
  static constexpr int NT = WAVES_M * WAVES_N * 64;
  static constexpr int BM = TH * TW;
  static constexpr int PAD = KS / 2;
  static constexpr int PH = (TH - 1) * STRIDE + KS;
  static constexpr int PW = (TW - 1) * STRIDE + KS;
  static constexpr int PS = KC + 4;
  static constexpr int TAPS = KS * KS;
  static constexpr int KV = KC / 4;
  static constexpr int A_FLOATS = PH * PW * PS;
  static constexpr int B_FLOATS = TAPS * BN * PS;
  static constexpr size_t LDS_BYTES = (size_t)(A_FLOATS + B_FLOATS) * sizeof(float);
  static_assert(BM == WAVES_M * WTM * 32, "pixel tile must match the wave layout");
  static_assert(BN == WAVES_N * WTN * 32, "channel tile must match the wave layout");
  static_assert(KC % 8 == 0, "KC must cover whole b128 operand pairs");
};

template <int KS, int STRIDE, int TH, int TW, int BN, int KC, int WAVES_M, int WAVES_N,
          int WTM, int WTN>
__global__ void __launch_bounds__(WAVES_M* WAVES_N * 64, 2)
conv_mfma_kernel(const ConvArgs a) {
  using T = ConvTile<KS, STRIDE, TH, TW, BN, KC, WAVES_M, WAVES_N, WTM, WTN>;
  constexpr int NT = T::NT, PW = T::PW, PH = T::PH, PS = T::PS, TAPS = T::TAPS, KV = T::KV;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + T::A_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_m = wave / WAVES_N;
  const int wave_n = wave % WAVES_N;
  const int li = lane & 31;
  const int lh = lane >> 5;

  int bid = blockIdx.x;
  const int tile_x = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int tile_y = bid % a.tiles_y;
  const int img = bid / a.tiles_y;
  const int cb = blockIdx.y;
  const int oy0 = tile_y * TH, ox0 = tile_x * TW;
  const int iy0 = oy0 * STRIDE - T::PAD, ix0 = ox0 * STRIDE - T::PAD;

  int a_off[WTM], b_off[WTN];
#pragma unroll
  for (int wm = 0; wm < WTM; ++wm) {
    const int m = (wave_m * WTM + wm) * 32 + li;
    a_off[wm] = ((m / TW) * STRIDE * PW + (m % TW) * STRIDE) * PS + 4 * lh;
  }
#pragma unroll
  for (int wn = 0; wn < WTN; ++wn) {
    const int n = (wave_n * WTN + wn) * 32 + li;
    b_off[wn] = n * PS + 4 * lh;
  }

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int wm = 0; wm < WTM; ++wm)
#pragma unroll
    for (int wn = 0; wn < WTN; ++wn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[wm][wn][r] = 0.f;

  const float* wblk = a.wpk + (size_t)cb * a.nchunks * (TAPS * BN * KC);

  for (int ch = 0; ch < a.nchunks; ++ch) {
    const int cbeg = ch * KC;
    __syncthreads();  // everyone is done reading the previous chunk
    {
      // ---- stage the halo patch of this channel chunk
      const bool from0 = cbeg < a.c0;
      const float* src = from0 ? a.src0 : a.src1;
      const int cs = from0 ? cbeg : cbeg - a.c0;
      const int ld = from0 ? a.ld0 : a.ld1;
      const int up = from0 ? a.up0 : 0;
      const int cvalid = (from0 ? a.c0 : a.c1) - cs;  // channels left in this source
      const int vec = from0 ? a.vec0 : a.vec1;
      const int hs = up ? (a.h_in >> 1) : a.h_in;
      const int ws = up ? (a.w_in >> 1) : a.w_in;
      for (int idx = tid; idx < PH * PW * KV; idx += NT) {
        const int p = idx / KV, q = idx % KV;
        const int iy = iy0 + p / PW, ix = ix0 + p % PW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < a.h_in && ix >= 0 && ix < a.w_in) {
          const int sy = up ? (iy >> 1) : iy, sx = up ? (ix >> 1) : ix;
          const float* gp = src + ((size_t)(img * hs + sy) * ws + sx) * ld + cs + 4 * q;
          if (vec && 4 * q + 4 <= cvalid) {
            v = *reinterpret_cast<const f32x4*>(gp);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (4 * q + e < cvalid) v[e] = gp[e];
          }
        }
        *reinterpret_cast<f32x4*>(&As[p * PS + 4 * q]) = v;
      }
      // ---- stage the weights of this chunk (pre-packed, contiguous)
      const f32x4* wsrc = reinterpret_cast<const f32x4*>(wblk + (size_t)ch * (TAPS * BN * KC));
      for (int idx = tid; idx < TAPS * BN * KV; idx += NT) {
        const int row = idx / KV, q = idx % KV;
        *reinterpret_cast<f32x4*>(&Bs[row * PS + 4 * q]) = wsrc[idx];
      }
    }
    __syncthreads();

#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int toff = ((tap / KS) * PW + (tap % KS)) * PS;
#pragma unroll
      for (int s = 0; s < KC / 8; ++s) {
        f32x4 av[WTM], bv[WTN];
#pragma unroll
        for (int wm = 0; wm < WTM; ++wm)
          av[wm] = *reinterpret_cast<const f32x4*>(&As[a_off[wm] + toff + 8 * s]);
#pragma unroll
        for (int wn = 0; wn < WTN; ++wn)
          bv[wn] = *reinterpret_cast<const f32x4*>(&Bs[tap * BN * PS + b_off[wn] + 8 * s]);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int wm = 0; wm < WTM; ++wm)
#pragma unroll
            for (int wn = 0; wn < WTN; ++wn)
              acc[wm][wn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[wm][t], bv[wn][t],
                                                                  acc[wm][wn], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: C/D layout col = lane&31 (channel), row = (r&3)+8*(r>>2)+4*(lane>>5)
#pragma unroll
  for (int wn = 0; wn < WTN; ++wn) {
    const int co = cb * BN + (wave_n * WTN + wn) * 32 + li;
    const bool cok = co < a.c_out;
    const float sc = cok ? a.scale[co] : 0.f;
    const float sh = cok ? a.shift[co] : 0.f;
#pragma unroll
    for (int wm = 0; wm < WTM; ++wm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int m = (wave_m * WTM + wm) * 32 + row;
        const int oy = oy0 + m / TW, ox = ox0 + m % TW;
        if (cok && oy < a.h_out && ox < a.w_out) {
          float v = acc[wm][wn][r] * sc + sh;
          if (a.relu) v = fmaxf(v, 0.f);
          a.out[((size_t)(img * a.h_out + oy) * a.w_out + ox) * a.ldo + co] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// tile configurations
// ---------------------------------------------------------------------------
enum CfgId { CFG_3A, CFG_3B, CFG_3C, CFG_3S2, CFG_1A, CFG_1B, CFG_1C };

struct Cfg {
  CfgId id;
  int th, tw, bn, kc, taps;
};

Cfg select_cfg(const dn_conv_desc& d) {
  if (d.ksize == 3) {
    if (d.stride == 2) return {CFG_3S2, 8, 16, 64, 8, 9};
    if (d.c_out <= 32) return {CFG_3A, 8, 32, 32, 16, 9};
    if (d.c_out <= 64) return {CFG_3B, 8, 32, 64, 16, 9};
    return {CFG_3C, 8, 16, 128, 8, 9};
  }
  if (d.c_out <= 32) return {CFG_1A, 8, 32, 32, 32, 1};
  if (d.c_out <= 64) return {CFG_1B, 8, 32, 64, 32, 1};
  return {CFG_1C, 8, 16, 128, 32, 1};
}

int validate(const dn_conv_desc* d) {
  DN_REQUIRE(d != nullptr, "conv: null descriptor");
  DN_REQUIRE(d->ksize == 1 || d->ksize == 3, "conv: ksize %d unsupported (1 or 3)", d->ksize);
  DN_REQUIRE(d->stride == 1 || (d->stride == 2 && d->ksize == 3),
             "conv: stride %d with ksize %d unsupported", d->stride, d->ksize);
  DN_REQUIRE(d->n_images > 0 && d->h_in > 0 && d->w_in > 0, "conv: empty input");
  DN_REQUIRE(d->c0 > 0 && d->c1 >= 0 && d->c_out > 0, "conv: bad channel counts");
  DN_REQUIRE(d->ld0 >= d->c0 && (d->c1 == 0 || d->ld1 >= d->c1) && d->ldo >= d->c_out,
             "conv: pixel strides smaller than channel counts");
  DN_REQUIRE(!d->up0 || (d->h_in % 2 == 0 && d->w_in % 2 == 0),
             "conv: x2-upsampled source needs even h_in/w_in");
  const Cfg c = select_cfg(*d);
  DN_REQUIRE(d->c1 == 0 || d->c0 % c.kc == 0,
             "conv: concat needs c0 (%d) to be a multiple of the chunk (%d)", d->c0, c.kc);
  return DN_OK;
}

int nchunks_of(const dn_conv_desc& d, const Cfg& c) { return (d.c0 + d.c1 + c.kc - 1) / c.kc; }
int ncb_of(const dn_conv_desc& d, const Cfg& c) { return (d.c_out + c.bn - 1) / c.bn; }

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wpk,
                                    int c_out, int c_in, int ks, int bn, int kc, int nchunks,
                                    long total) {
  const int taps = ks * ks;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    long r = idx;
    const int k = r % kc; r /= kc;
    const int n = r % bn; r /= bn;
    const int tap = r % taps; r /= taps;
    const int ch = r % nchunks;
    const int cb = r / nchunks;
    const int co = cb * bn + n, ci = ch * kc + k;
    float v = 0.f;
    if (co < c_out && ci < c_in) v = w[((size_t)co * c_in + ci) * taps + tap];
    wpk[idx] = v;
  }
}

__global__ void fold_bn_kernel(const float* bias, const float* gamma, const float* beta,
                               const float* mean, const float* var, float eps, int n,
                               float* scale, float* shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float b = bias ? bias[i] : 0.f;
  if (gamma) {
    const float sc = gamma[i] / sqrtf(var[i] + eps);
    scale[i] = sc;
    shift[i] = (b - mean[i]) * sc + beta[i];
  } else {
    scale[i] = 1.f;
    shift[i] = b;
  }
}

template <int KS, int STRIDE, int TH, int TW, int BN, int KC, int WAVES_M, int WAVES_N,
          int WTM, int WTN>
int launch(const ConvArgs& a, int ncb, hipStream_t stream) {
  using T = ConvTile<KS, STRIDE, TH, TW, BN, KC, WAVES_M, WAVES_N, WTM, WTN>;
  auto kern = conv_mfma_kernel<KS, STRIDE, TH, TW, BN, KC, WAVES_M, WAVES_N, WTM, WTN>;
  static_assert(T::LDS_BYTES <= 160 * 1024, "tile does not fit the 160 KiB LDS");
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)T::LDS_BYTES);
  if (e != hipSuccess)
    return dn::fail(DN_ERR_LAUNCH, "conv: hipFuncSetAttribute(%zu B LDS): %s", T::LDS_BYTES,
                    hipGetErrorString(e));
  dim3 grid((unsigned)(a.n_images * a.tiles_y * a.tiles_x), (unsigned)ncb);
  hipLaunchKernelGGL(kern, grid, dim3(T::NT), T::LDS_BYTES, stream, a);
  return dn::check_launch("conv_mfma_kernel");
}

}  // namespace

extern "C" size_t dn_conv_packed_weight_floats(const dn_conv_desc* d) {
  if (validate(d) != DN_OK) return 0;
  const Cfg c = select_cfg(*d);
  return (size_t)ncb_of(*d, c) * nchunks_of(*d, c) * c.taps * c.bn * c.kc;
}

extern "C" int dn_conv_pack_weights(const dn_conv_desc* d, const float* weight_oihw,
                                    float* packed, void* stream) {
  if (int rc = validate(d)) return rc;
  DN_REQUIRE(weight_oihw && packed, "conv pack: null pointer");
  const Cfg c = select_cfg(*d);
  const long total = (long)dn_conv_packed_weight_floats(d);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     weight_oihw, packed, d->c_out, d->c0 + d->c1, d->ksize, c.bn, c.kc,
                     nchunks_of(*d, c), total);
  return dn::check_launch("pack_weights_kernel");
}

extern "C" int dn_fold_bn(const float* bias, const float* gamma, const float* beta,
                          const float* mean, const float* var, float eps, int channels,
                          float* scale, float* shift, void* stream) {
  DN_REQUIRE(channels > 0 && scale && shift, "fold_bn: bad arguments");
  DN_REQUIRE(!gamma || (beta && mean && var), "fold_bn: gamma without beta/mean/var");
  hipLaunchKernelGGL(fold_bn_kernel, dim3((channels + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, bias, gamma, beta, mean, var, eps, channels, scale,
                     shift);
  return dn::check_launch("fold_bn_kernel");
}

extern "C" int dn_conv2d(const dn_conv_desc* d, const float* src0, const float* src1,
                         const float* packed, const float* scale, const float* shift,
                         float* out, void* stream) {
  if (int rc = validate(d)) return rc;
  DN_REQUIRE(src0 && packed && scale && shift && out, "conv: null pointer");
  DN_REQUIRE(d->c1 == 0 || src1, "conv: c1 > 0 but src1 is null");
  const Cfg c = select_cfg(*d);
  ConvArgs a;
  a.src0 = src0; a.src1 = src1; a.wpk = packed; a.scale = scale; a.shift = shift; a.out = out;
  a.n_images = d->n_images; a.h_in = d->h_in; a.w_in = d->w_in;
  const int pad = d->ksize / 2;
  a.h_out = (d->h_in + 2 * pad - d->ksize) / d->stride + 1;
  a.w_out = (d->w_in + 2 * pad - d->ksize) / d->stride + 1;
  a.c0 = d->c0; a.c1 = d->c1; a.up0 = d->up0; a.c_out = d->c_out; a.relu = d->relu;
  a.ld0 = d->ld0; a.ld1 = d->ld1; a.ldo = d->ldo;
  a.nchunks = nchunks_of(*d, c);
  a.tiles_x = (a.w_out + c.tw - 1) / c.tw;
  a.tiles_y = (a.h_out + c.th - 1) / c.th;
  auto aligned16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  a.vec0 = (d->c0 % 4 == 0 && d->ld0 % 4 == 0 && aligned16(src0)) ? 1 : 0;
  a.vec1 = (d->c1 > 0 && d->c1 % 4 == 0 && d->ld1 % 4 == 0 && aligned16(src1)) ? 1 : 0;
  DN_REQUIRE(aligned16(packed), "conv: packed weights must be 16-byte aligned");
  const int ncb = ncb_of(*d, c);
  hipStream_t s = (hipStream_t)stream;
  switch (c.id) {
    case CFG_3A:  return launch<3, 1, 8, 32, 32, 16, 4, 1, 2, 1>(a, ncb, s);
    case CFG_3B:  return launch<3, 1, 8, 32, 64, 16, 4, 1, 2, 2>(a, ncb, s);
    case CFG_3C:  return launch<3, 1, 8, 16, 128, 8, 2, 2, 2, 2>(a, ncb, s);
    case CFG_3S2: return launch<3, 2, 8, 16, 64, 8, 2, 2, 2, 1>(a, ncb, s);
    case CFG_1A:  return launch<1, 1, 8, 32, 32, 32, 4, 1, 2, 1>(a, ncb, s);
    case CFG_1B:  return launch<1, 1, 8, 32, 64, 32, 4, 1, 2, 2>(a, ncb, s);
    case CFG_1C:  return launch<1, 1, 8, 16, 128, 32, 2, 2, 2, 2>(a, ncb, s);
  }
  return dn::fail(DN_ERR_UNSUPPORTED, "conv: no tile configuration");
}
