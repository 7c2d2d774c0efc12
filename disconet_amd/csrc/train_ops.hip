// Training-step kernels around the convs (include/disconet_train.h): batch-norm in
// training mode and its backward, bias / channel sums, the training form of the
// DiscoGraph fusion tail, the detection loss and Adam.  All HBM-bound row-major
// streaming over NHWC maps: lanes run along the channel axis (coalesced rows),
// per-channel sums are kept in double (the batch variance is E[z^2] - mean^2 over
// up to 1.3 M pixels) in LDS per workgroup and merged with one double atomic per
// (workgroup, channel).
//
// Replaces the autograd graph of upstream:coperception/utils/CoDetModule.py :: step
// (nn.BatchNorm2d / BatchNorm3d in train(), F.relu, torch.exp / div / mul of the fusion
// loop in upstream:.../det/DiscoNet.py :: forward, loss.py's focal / smooth-L1 losses,
// torch.optim.Adam).
#ifndef DN_REDUCE_REVERSE
#define DN_REDUCE_REVERSE 0      // tools/ab: 1 = the per-channel reductions walk the map from its end (profiles/r06_reduce_reverse_ab.txt)
#endif
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <cmath>
#include <initializer_list>
#include <type_traits>

#include "disconet_train.h"
#include "dn_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMaxC = 512;      // channels of the widest map on the path (conv4_*)

__device__ inline int lanes_for(int c) {   // power-of-two channel lanes per row, <= 64
  int t = 1;
  while (t < c && t < 64) t <<= 1;
  return t;
}

// loss VALUES only (det_loss_kernel, kd_kl_kernel): one f64 atomic per workgroup; the reported scalar may differ in
// its last bits between runs, no gradient or parameter depends on it
__device__ inline void atomic_add_f64(double* p, double v) { atomicAdd(p, v); }

// ---------------------------------------------------------------------------------
// per-(group, channel) sums of up to two quantities over the rows of a group -- DETERMINISTIC: the rows a thread
// adds, the order in which a workgroup adds its threads' partials (LDS, by row-lane index) and the order in which
// the workgroups' partials are added (fold_partials_kernel, by block index) are all fixed by the launch shape,
// so two runs of a training step produce the same bits (round 2 used f64 atomics in LDS and in global memory).
// ---------------------------------------------------------------------------------
// F(row_in_group, c, &q0, &q1) yields the two addends; grid = (blocks per group, groups).  The workgroup's partial
// goes to part_g[blockIdx.x * NQ * c + q * c + channel].
constexpr int kRedSlots = 2048;     // ty_n * c <= 2048 for every launch shape (c <= kMaxC)
template <int NQ, class F>
__device__ inline void group_channel_sums(int c, long rows_per_group, double* part_g, F f) {
  __shared__ double red[2][kRedSlots];
  const int tx_n = lanes_for(c), ty_n = blockDim.x / tx_n;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const long chunk = (rows_per_group + gridDim.x - 1) / gridDim.x;
  const long r0 = blockIdx.x * chunk;
  const long r1 = r0 + chunk < rows_per_group ? r0 + chunk : rows_per_group;
  for (int cc = tx; cc < c; cc += tx_n) {
    double s0 = 0.0, s1 = 0.0;
    for (long r = r0 + ty; r < r1; r += ty_n) {
      float q0 = 0.f, q1 = 0.f;
      f(r, cc, q0, q1);
      s0 += q0;
      if (NQ > 1) s1 += q1;
    }
    red[0][ty * c + cc] = s0;
    if (NQ > 1) red[1][ty * c + cc] = s1;
  }
  __syncthreads();
  double* out = part_g + (size_t)blockIdx.x * NQ * c;
  for (int i = threadIdx.x; i < NQ * c; i += blockDim.x) {
    const int q = i / c, cc = i % c;
    double t = 0.0;
    for (int y = 0; y < ty_n; ++y) t += red[q][y * c + cc];
    out[i] = t;
  }
}

// float4 form: lanes over groups of 4 channels (c % 4 == 0, 16-byte aligned rows).
// U: rows a thread fetches before it adds them.  The adds stay in row order (the same sums, bit for bit, as U = 1), but U rows'
// loads are in flight at once: with one row per iteration the kernels moved 3.8-4.5 TB/s, bound by bytes in flight per CU
// (16 waves x 64 lanes x 2 loads of 16 B against ~2 us of HBM latency), not by the HBM (round 5, profiles/r05_train_pmc.txt).
template <int NQ, int U = 4, class F>
__device__ inline void group_channel_sums_v4(int c, long rows_per_group, double* part_g, F f) {
  __shared__ double red[2][kRedSlots];
  const int c4n = c >> 2;
  const int tx_n = lanes_for(c4n), ty_n = blockDim.x / tx_n;
  const int tx = threadIdx.x % tx_n, ty = threadIdx.x / tx_n;
  const long chunk = (rows_per_group + gridDim.x - 1) / gridDim.x;
#if DN_REDUCE_REVERSE
  // tools/ab: the workgroups take their row chunks from the END of the map first (what the producer wrote last), same partial slots
  const unsigned lb = gridDim.x - 1 - blockIdx.x;
#else
  const unsigned lb = blockIdx.x;
#endif
  const long r0 = lb * chunk;
  const long r1 = r0 + chunk < rows_per_group ? r0 + chunk : rows_per_group;
  for (int c4 = tx; c4 < c4n; c4 += tx_n) {
    double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
    long r = r0 + ty;
    if constexpr (U > 1) {
      for (; r + (long)(U - 1) * ty_n < r1; r += (long)U * ty_n) {
        f32x4 q0[U], q1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          q1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          f(r + (long)u * ty_n, c4, q0[u], q1[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s0[e] += q0[u][e];
            if (NQ > 1) s1[e] += q1[u][e];
          }
      }
    }
    for (; r < r1; r += ty_n) {
      f32x4 q0, q1 = {0.f, 0.f, 0.f, 0.f};
      f(r, c4, q0, q1);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s0[e] += q0[e];
        if (NQ > 1) s1[e] += q1[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][ty * c + 4 * c4 + e] = s0[e];
      if (NQ > 1) red[1][ty * c + 4 * c4 + e] = s1[e];
    }
  }
  __syncthreads();
  double* out = part_g + (size_t)lb * NQ * c;
  for (int i = threadIdx.x; i < NQ * c; i += blockDim.x) {
    const int q = i / c, cc = i % c;
    double t = 0.0;
    for (int y = 0; y < ty_n; ++y) t += red[q][y * c + cc];
    out[i] = t;
  }
}

// sums[g][i] = sum over the group's blocks of part[g][blk][i]   (i < per = NQ * c), in a FIXED order: one wavefront
// per output; lane l adds blocks l, l + 64, l + 128, ... in that order, then the 64 lane sums go through a xor
// butterfly (a fixed tree).  (A single thread walking up to 2048 partials was a 1 ms serial chain per reduction.)
__global__ void __launch_bounds__(256) fold_partials_kernel(const double* __restrict__ part, int n_blocks, int per,
                                                            int n_groups, double* __restrict__ sums) {
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (idx >= n_groups * per) return;
  const int g = idx / per, i = idx % per;
  const double* p = part + (size_t)g * n_blocks * per + i;
  double t = 0.0;
  for (int b = lane; b < n_blocks; b += 64) t += p[(size_t)b * per];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
  if (lane == 0) sums[idx] = t;
}

__device__ inline f32x4 ldv4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ inline f32x4 rstd4(const float* var, float eps) {
  const f32x4 v = ldv4(var);
  return f32x4{1.f / sqrtf(v[0] + eps), 1.f / sqrtf(v[1] + eps), 1.f / sqrtf(v[2] + eps),
               1.f / sqrtf(v[3] + eps)};
}

template <int U>
__global__ void __launch_bounds__(256)
bn_stats_v4_kernel(const float* __restrict__ z, long rows_per_group, int c, int ldz,
                   double* __restrict__ sums) {
  const int g = blockIdx.y;
  const float* zg = z + (size_t)g * rows_per_group * ldz;
  group_channel_sums_v4<2, U>(c, rows_per_group, sums + (size_t)g * gridDim.x * 2 * c,
                           [&](long r, int c4, f32x4& q0, f32x4& q1) {
                             q0 = ldv4(zg + r * ldz + 4 * c4);
                             q1 = q0 * q0;
                           });
}

__global__ void __launch_bounds__(256)
bn_apply_v4_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                   const float* __restrict__ var, const float* __restrict__ gamma,
                   const float* __restrict__ beta, float eps, int relu, long rows_per_group, int c,
                   int ldz, long total4, float* __restrict__ y, unsigned char* __restrict__ relu_mask) {
  const int c4n = c >> 2;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total4;
       idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / c4n;
    const int c4 = (int)(idx % c4n);
    const int g = (int)(row / rows_per_group);
    const f32x4 rs = rstd4(var + g * c + 4 * c4, eps);
    f32x4 v = (ldv4(z + row * ldz + 4 * c4) - ldv4(mean + g * c + 4 * c4)) * rs * ldv4(gamma + 4 * c4) +
              ldv4(beta + 4 * c4);
    // one byte per four channels: bit e = (y[4 c4 + e] > 0), what the backward's ReLU gate asks of y -- 1/16 of y's bytes
    if (relu_mask) relu_mask[idx] = (unsigned char)((v[0] > 0.f) | ((v[1] > 0.f) << 1) | ((v[2] > 0.f) << 2) | ((v[3] > 0.f) << 3));
    if (relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    *reinterpret_cast<f32x4*>(y + row * (long)c + 4 * c4) = v;
  }
}

// One group, c / 4 a power of two, fewer than 2^31 float4s (every BatchNorm of the conv stack): the general kernel above spends
// ~300 VALU instructions per float4 on two 64-bit divisions and four 1 / sqrtf -- it is instruction-bound at ~4.8 TB/s, not
// HBM-bound (round 5: rocprofv3 counters, profiles/r05_train_pmc.txt).  Here 1 / sqrtf(var + eps) is computed once per workgroup
// into LDS (the same expression: the same bits) and (row, channel quad) come from a shift and a mask.  DN_BN_LEGACY=1 routes
// every call to the general kernels (tests: bitwise A/B).
template <bool SP>
__global__ void __launch_bounds__(256)
bn_apply_v4_fast_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ var,
                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int relu, int c, int sh,
                        int ldz, unsigned total4, float* __restrict__ y, unsigned char* __restrict__ relu_mask,
                        unsigned char* __restrict__ y_sp, unsigned hw, unsigned* __restrict__ flags) {
  __shared__ __attribute__((aligned(16))) float rstd_s[kMaxC];
  for (int i = threadIdx.x; i < c; i += 256) rstd_s[i] = 1.f / sqrtf(var[i] + eps);
  __syncthreads();
  const unsigned c4m = (1u << sh) - 1u;
  float amax = 0.f;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total4; idx += gridDim.x * 256u) {
    const unsigned row = idx >> sh;
    const int c4 = (int)(idx & c4m);
    const f32x4 rs = *reinterpret_cast<const f32x4*>(&rstd_s[4 * c4]);
    f32x4 v = (ldv4(z + (size_t)row * ldz + 4 * c4) - ldv4(mean + 4 * c4)) * rs * ldv4(gamma + 4 * c4) +
              ldv4(beta + 4 * c4);
    if (relu_mask) relu_mask[idx] = (unsigned char)((v[0] > 0.f) | ((v[1] > 0.f) << 1) | ((v[2] > 0.f) << 2) | ((v[3] > 0.f) << 3));
    if (relu) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    *reinterpret_cast<f32x4*>(y + (size_t)row * c + 4 * c4) = v;
    if constexpr (SP) {
      // the SP copy of y (the next layer's forward operand on the split-f16 engine): the piece assembly of
      // bn_bwd_apply_v4_kernel<true> -- two adjacent threads hold the 8 channels of a piece and swap halves
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      typedef unsigned u2 __attribute__((ext_vector_type(2)));
      f32x4 x;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x[e] = fminf(fmaxf(v[e], -65504.f), 65504.f);
        amax = fmaxf(amax, fabsf(x[e]));
      }
      const h4 hi = __builtin_convertvector(x, h4);
      const h4 lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x4), h4);
      const u2 hu = __builtin_bit_cast(u2, hi), lu = __builtin_bit_cast(u2, lo);
      const bool odd = c4 & 1;
      const unsigned t0 = __shfl_xor(odd ? hu[0] : lu[0], 1), t1 = __shfl_xor(odd ? hu[1] : lu[1], 1);
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      const u4 piece = odd ? u4{t0, t1, lu[0], lu[1]} : u4{hu[0], hu[1], t0, t1};
      const unsigned img = row / hw, px = row - img * hw;
      const int oct8 = c4 >> 1, cg = oct8 >> 1, oct = oct8 & 1, cg_total = c >> 4;
      const size_t q = ((size_t)img * cg_total + cg) * 4 + (odd ? 2 : 0) + oct;
      *reinterpret_cast<u4*>(y_sp + (q * hw + px) * 16) = piece;
    }
  }
  if constexpr (SP) {
    if (amax > 16384.f && flags) atomicOr(flags, amax >= 65504.f ? 3u : 2u);
  }
}

__global__ void __launch_bounds__(256)
bn_stats_kernel(const float* __restrict__ z, long rows_per_group, int c, int ldz,
                double* __restrict__ sums) {
  const int g = blockIdx.y;
  const float* zg = z + (size_t)g * rows_per_group * ldz;
  group_channel_sums<2>(c, rows_per_group, sums + (size_t)g * gridDim.x * 2 * c,
                        [&](long r, int cc, float& q0, float& q1) {
                          const float v = zg[r * ldz + cc];
                          q0 = v;
                          q1 = v * v;
                        });
}

__global__ void bn_stats_finalize_kernel(const double* __restrict__ sums, int n, int c, long rows,
                                         float* __restrict__ mean, float* __restrict__ var) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (group, channel)
  if (i >= n) return;
  const int g = i / c, cc = i % c;
  const double m = sums[(size_t)g * 2 * c + cc] / rows;
  const double v = sums[(size_t)g * 2 * c + c + cc] / rows - m * m;
  mean[i] = (float)m;
  var[i] = (float)(v > 0.0 ? v : 0.0);
}

__global__ void __launch_bounds__(256)
bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                const float* __restrict__ var, const float* __restrict__ gamma,
                const float* __restrict__ beta, float eps, int relu, long rows_per_group, int c,
                int ldz, long total, float* __restrict__ y) {
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / c;
    const int cc = (int)(idx % c);
    const int g = (int)(row / rows_per_group);
    const float rstd = 1.f / sqrtf(var[g * c + cc] + eps);
    float v = (z[row * ldz + cc] - mean[g * c + cc]) * rstd * gamma[cc] + beta[cc];
    if (relu) v = fmaxf(v, 0.f);
    y[idx] = v;
  }
}

__global__ void bn_update_running_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                         int n_groups, long rows, int c, const int* __restrict__ order,
                                         float momentum, float* __restrict__ rmean,
                                         float* __restrict__ rvar) {
  const int cc = blockIdx.x * blockDim.x + threadIdx.x;
  if (cc >= c) return;
  float rm = rmean[cc], rv = rvar[cc];
  const float unbias = rows > 1 ? (float)rows / (float)(rows - 1) : 1.f;
  // the recurrence runs in group order; the groups' statistics are fetched eight at a time (one round trip instead of eight)
  for (int k0 = 0; k0 < n_groups; k0 += 8) {
    float m[8], v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u < n_groups ? k0 + u : n_groups - 1;
      const int g = order ? order[k] : k;
      m[u] = mean[g * c + cc];
      v[u] = var[g * c + cc];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + u < n_groups) {
        rm = (1.f - momentum) * rm + momentum * m[u];
        rv = (1.f - momentum) * rv + momentum * (v[u] * unbias);
      }
  }
  rmean[cc] = rm;
  rvar[cc] = rv;
}

// ---- fold + finish in ONE launch (round 6: the step ran ~75 fold_partials launches of 6.7 us each plus as many 3-5 us finalize
// launches behind them).  One wavefront per (group, channel) folds BOTH quantities of its channel in fold_partials_kernel's order
// (lane l adds blocks l, l + 64, ...; xor butterfly): the same sums bit for bit, then the finish arithmetic of the kernel it
// replaces -- bn_stats_finalize_kernel (+ optionally bn_update_running_kernel: one group), bn_param_grad_kernel (one group),
// channel_sum_finalize_kernel.
// DN_FOLD_BATCH: partials a lane fetches before it adds them (in the same order: the same bits).  With one load per iteration a
// lane's up to 16 partials were 16 L2 round trips in a row: 7-11 us per launch, 72 launches per step (profiles/r06_fold_batch_ab.txt).
#ifndef DN_FOLD_BATCH
#define DN_FOLD_BATCH 8
#endif
__device__ inline double fold_one(const double* __restrict__ p, int n_blocks, size_t stride, int lane) {
  double t = 0.0;
  int b = lane;
  if constexpr (DN_FOLD_BATCH > 1) {
    for (; b + 64 * (DN_FOLD_BATCH - 1) < n_blocks; b += 64 * DN_FOLD_BATCH) {
      double v[DN_FOLD_BATCH];
#pragma unroll
      for (int u = 0; u < DN_FOLD_BATCH; ++u) v[u] = p[(size_t)(b + 64 * u) * stride];
#pragma unroll
      for (int u = 0; u < DN_FOLD_BATCH; ++u) t += v[u];
    }
  }
  for (; b < n_blocks; b += 64) t += p[(size_t)b * stride];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
  return t;
}
// two quantities of one channel (p0, p1: the same blocks): both fetches of a batch in flight together, each sum in fold_one's order
__device__ inline void fold_two(const double* __restrict__ p0, const double* __restrict__ p1, int n_blocks, size_t stride, int lane,
                                double& t0, double& t1) {
  double a = 0.0, c = 0.0;
  int b = lane;
  if constexpr (DN_FOLD_BATCH > 1) {
    for (; b + 64 * (DN_FOLD_BATCH - 1) < n_blocks; b += 64 * DN_FOLD_BATCH) {
      double v[DN_FOLD_BATCH], w[DN_FOLD_BATCH];
#pragma unroll
      for (int u = 0; u < DN_FOLD_BATCH; ++u) {
        v[u] = p0[(size_t)(b + 64 * u) * stride];
        w[u] = p1[(size_t)(b + 64 * u) * stride];
      }
#pragma unroll
      for (int u = 0; u < DN_FOLD_BATCH; ++u) {
        a += v[u];
        c += w[u];
      }
    }
  }
  for (; b < n_blocks; b += 64) {
    const double v = p0[(size_t)b * stride], w = p1[(size_t)b * stride];
    a += v;
    c += w;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    c += __shfl_xor(c, o, 64);
  }
  t0 = a;
  t1 = c;
}
__global__ void __launch_bounds__(256)
fold_stats_finish_kernel(const double* __restrict__ part, int n_blocks, int c, int n_groups, long norm_rows, double* __restrict__ sums,
                         float* __restrict__ mean, float* __restrict__ var, float* __restrict__ rmean, float* __restrict__ rvar,
                         float momentum, long unbias_rows) {
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (idx >= n_groups * c) return;
  const int g = idx / c, cc = idx % c;
  const double* p = part + (size_t)g * n_blocks * 2 * c + cc;
  double t0, t1;
  fold_two(p, p + c, n_blocks, 2 * c, lane, t0, t1);
  if (lane == 0) {
    sums[(size_t)g * 2 * c + cc] = t0;
    sums[(size_t)g * 2 * c + c + cc] = t1;
    const double m = t0 / norm_rows;
    const double v = t1 / norm_rows - m * m;
    const float mf = (float)m, vf = (float)(v > 0.0 ? v : 0.0);
    mean[idx] = mf;
    var[idx] = vf;
    if (rmean) {      // (one group) bn_update_running_kernel's arithmetic
      const float unbias = unbias_rows > 1 ? (float)unbias_rows / (float)(unbias_rows - 1) : 1.f;
      rmean[cc] = (1.f - momentum) * rmean[cc] + momentum * mf;
      rvar[cc] = (1.f - momentum) * rvar[cc] + momentum * (vf * unbias);
    }
  }
}
__global__ void __launch_bounds__(256)
fold_param_grad_kernel(const double* __restrict__ part, int n_blocks, int c, double* __restrict__ sums, float* __restrict__ dgamma,
                       float* __restrict__ dbeta, int accumulate) {      // one group
  const int cc = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (cc >= c) return;
  double t0, t1;
  fold_two(part + cc, part + c + cc, n_blocks, 2 * c, lane, t0, t1);
  if (lane == 0) {
    sums[cc] = t0;
    sums[c + cc] = t1;
    double s1 = 0.0, s2 = 0.0;      // (bn_param_grad_kernel's sum over its one group)
    s1 += t0;
    s2 += t1;
    dbeta[cc] = (accumulate ? dbeta[cc] : 0.f) + (float)s1;
    dgamma[cc] = (accumulate ? dgamma[cc] : 0.f) + (float)s2;
  }
}
__global__ void __launch_bounds__(256)
fold_channel_sum_kernel(const double* __restrict__ part, int n_blocks, int c, double* __restrict__ sums, float* __restrict__ out,
                        int accumulate) {
  const int cc = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (cc >= c) return;
  const double t = fold_one(part + cc, n_blocks, c, lane);
  if (lane == 0) {
    sums[cc] = t;
    out[cc] = (accumulate ? out[cc] : 0.f) + (float)t;
  }
}

// ---- the channel-sum folds of a whole backward pass in ONE launch (dn_channel_sum_fold_multi).  The sums they finish -- bias
// gradients: leaves of the backward -- are not read before the optimizer step, so the launches that leave the partials
// (dn_bn_train_backward_finish_bias_deferred, dn_channel_sum_partial) need not be followed by a fold each (26 launches of
// 5 us per step).  Jobs ride in the kernel arguments; one wavefront per (job, channel) runs fold_channel_sum_kernel's body.
constexpr int kFoldJobs = 32;
struct FoldJobs {
  const double* part[kFoldJobs];
  double* sums[kFoldJobs];
  float* out[kFoldJobs];
  int n_blocks[kFoldJobs], c[kFoldJobs], accumulate[kFoldJobs];
  int first[kFoldJobs + 1];      // first wavefront of job j; first[n_jobs] = all wavefronts
};
__global__ void __launch_bounds__(256) fold_channel_sum_multi_kernel(const FoldJobs J, int n_jobs) {
  const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wv >= J.first[n_jobs]) return;
  int j = 0;
  while (j + 1 < n_jobs && J.first[j + 1] <= wv) ++j;      // wave-uniform
  const int cc = wv - J.first[j], c = J.c[j];
  const double t = fold_one(J.part[j] + cc, J.n_blocks[j], c, lane);
  if (lane == 0) {
    J.sums[j][cc] = t;
    float* out = J.out[j];
    out[cc] = (J.accumulate[j] ? out[cc] : 0.f) + (float)t;
  }
}

// incoming gradient of one element: dy_a (optionally the 2 x 2 block sum of a map at twice
// the resolution) + dy_b, gated by the ReLU
struct GradSrc {
  const float* dy_a;
  const float* dy_b;
  const float* y;
  int ld_a, up_a, ld_b, relu, h, w, c;
  __device__ inline float operator()(long row, int cc) const {
    float g;
    if (up_a == 2) {
      // dy_a is the space-to-depth image of the gradient: [img][h / 2][w / 2][4 c], pixel (y, x) channel cc at
      // (y / 2, x / 2), channel ((y & 1) * 2 + (x & 1)) * c + cc -- what ONE split-f16 launch over the four parity classes of a
      // stride-2 layer's data gradient writes (train.py :: _dgrad)
      const long img = row / ((long)h * w);
      const int p = (int)(row % ((long)h * w));
      const int py = p / w, px = p % w;
      g = dy_a[((img * (h >> 1) + (py >> 1)) * (long)(w >> 1) + (px >> 1)) * ld_a + ((py & 1) * 2 + (px & 1)) * c + cc];
    } else if (up_a) {
      const long img = row / ((long)h * w);
      const int p = (int)(row % ((long)h * w));
      const int py = p / w, px = p % w;
      const float* b = dy_a + ((img * 2 * h + 2 * py) * (2L * w) + 2 * px) * ld_a + cc;
      g = (b[0] + b[ld_a]) + (b[2L * w * ld_a] + b[(2L * w + 1) * ld_a]);
    } else {
      g = dy_a[row * ld_a + cc];
    }
    if (dy_b) g += dy_b[row * ld_b + cc];
    if (relu == 2) {          // y is the byte mask of dn_bn_train_apply_mask (c % 4 == 0)
      if (!((reinterpret_cast<const unsigned char*>(y)[(row * c + cc) >> 2] >> (cc & 3)) & 1)) g = 0.f;
    } else if (relu && !(y[row * c + cc] > 0.f)) g = 0.f;
    return g;
  }
  // v4 for rows below 2^31 (the one-group fast kernels): the same values, the pixel decode in 32-bit arithmetic
  __device__ inline f32x4 v4u(unsigned row, int c4) const {
    f32x4 g;
    if (up_a) {
      const unsigned hw = (unsigned)h * (unsigned)w;
      const unsigned img = row / hw, p = row - img * hw;
      const unsigned py = p / (unsigned)w, px = p - py * (unsigned)w;
      if (up_a == 2) {
        g = ldv4(dy_a + (((size_t)img * (h >> 1) + (py >> 1)) * (size_t)(w >> 1) + (px >> 1)) * ld_a + ((py & 1) * 2 + (px & 1)) * c + 4 * c4);
      } else {
        const float* b = dy_a + (((size_t)img * 2 * h + 2 * py) * (2L * w) + 2 * px) * ld_a + 4 * c4;
        g = (ldv4(b) + ldv4(b + ld_a)) + (ldv4(b + 2L * w * ld_a) + ldv4(b + (2L * w + 1) * ld_a));
      }
    } else {
      g = ldv4(dy_a + (size_t)row * ld_a + 4 * c4);
    }
    if (dy_b) g += ldv4(dy_b + (size_t)row * ld_b + 4 * c4);
    if (relu == 2) {
      const unsigned m = reinterpret_cast<const unsigned char*>(y)[(size_t)row * (c >> 2) + c4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (!((m >> e) & 1)) g[e] = 0.f;
    } else if (relu) {
      const f32x4 yv = ldv4(y + (size_t)row * c + 4 * c4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (!(yv[e] > 0.f)) g[e] = 0.f;
    }
    return g;
  }
  __device__ inline f32x4 v4(long row, int c4) const {
    f32x4 g;
    if (up_a == 2) {
      const long img = row / ((long)h * w);
      const int p = (int)(row % ((long)h * w));
      const int py = p / w, px = p % w;
      g = ldv4(dy_a + ((img * (h >> 1) + (py >> 1)) * (long)(w >> 1) + (px >> 1)) * ld_a + ((py & 1) * 2 + (px & 1)) * c + 4 * c4);
    } else if (up_a) {
      const long img = row / ((long)h * w);
      const int p = (int)(row % ((long)h * w));
      const int py = p / w, px = p % w;
      const float* b = dy_a + ((img * 2 * h + 2 * py) * (2L * w) + 2 * px) * ld_a + 4 * c4;
      g = (ldv4(b) + ldv4(b + ld_a)) + (ldv4(b + 2L * w * ld_a) + ldv4(b + (2L * w + 1) * ld_a));
    } else {
      g = ldv4(dy_a + row * ld_a + 4 * c4);
    }
    if (dy_b) g += ldv4(dy_b + row * ld_b + 4 * c4);
    if (relu == 2) {
      const unsigned m = reinterpret_cast<const unsigned char*>(y)[row * (long)(c >> 2) + c4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (!((m >> e) & 1)) g[e] = 0.f;
    } else if (relu) {
      const f32x4 yv = ldv4(y + row * (long)c + 4 * c4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (!(yv[e] > 0.f)) g[e] = 0.f;
    }
    return g;
  }
};

template <int U>
__global__ void __launch_bounds__(256)
bn_bwd_reduce_v4_kernel(GradSrc src, const float* __restrict__ z, const float* __restrict__ mean,
                        const float* __restrict__ var, float eps, long rows_per_group,
                        double* __restrict__ sums) {
  const int g = blockIdx.y, c = src.c;
  const long base = (long)g * rows_per_group;
  group_channel_sums_v4<2, U>(c, rows_per_group, sums + (size_t)g * gridDim.x * 2 * c,
                           [&](long r, int c4, f32x4& q0, f32x4& q1) {
                             q0 = src.v4(base + r, c4);
                             const f32x4 zh = (ldv4(z + (base + r) * c + 4 * c4) - ldv4(mean + g * c + 4 * c4)) *
                                              rstd4(var + g * c + 4 * c4, eps);
                             q1 = q0 * zh;
                           });
}

// SP: a second copy of dz, multiplied by the power of two `sp_lift`, as the split-planar f16 hi / lo tensor of the inference conv
// engine (include/disconet_hip.h "SP tensor": [image][c / 16][4 quarters][h][w] x 16 bytes, quarter = 2 * part + octet) -- the
// operand of the split-f16 data gradient (dn_spconv2d_nhwc).  A piece is 8 channels of one pixel = the values of TWO adjacent
// threads (c % 16 == 0: an even / odd pair never straddles a wavefront or the end of the loop): they swap halves, the even thread
// writes the hi piece, the odd one the lo piece.  The split is dn_sp_from_nhwc's (clamp to +-65504, hi = half(x), lo = half(x - hi)),
// magnitudes reported into the conv engine's sticky range word (`flags`).
template <bool SP>
__global__ void __launch_bounds__(256)
bn_bwd_apply_v4_kernel(GradSrc src, const float* __restrict__ z, const float* __restrict__ mean,
                       const float* __restrict__ var, const float* __restrict__ gamma, float eps,
                       long rows_per_group, long norm_rows, const double* __restrict__ sums, long total4,
                       float* __restrict__ dz, unsigned char* __restrict__ dz_sp, float sp_lift, unsigned hw,
                       unsigned* __restrict__ flags) {
  const int c = src.c, c4n = c >> 2;
  float amax = 0.f;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total4;
       idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / c4n;
    const int c4 = (int)(idx % c4n);
    const int g = (int)(row / rows_per_group);
    const f32x4 rs = rstd4(var + g * c + 4 * c4, eps);
    const f32x4 zh = (ldv4(z + row * c + 4 * c4) - ldv4(mean + g * c + 4 * c4)) * rs;
    const double* sg = sums + (size_t)g * 2 * c + 4 * c4;
    f32x4 m1, m2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      m1[e] = (float)(sg[e] / norm_rows);
      m2[e] = (float)(sg[c + e] / norm_rows);
    }
    const f32x4 v = ldv4(gamma + 4 * c4) * rs * (src.v4(row, c4) - m1 - zh * m2);
    *reinterpret_cast<f32x4*>(dz + row * c + 4 * c4) = v;
    if constexpr (SP) {
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      typedef unsigned u2 __attribute__((ext_vector_type(2)));
      f32x4 x;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x[e] = fminf(fmaxf(v[e] * sp_lift, -65504.f), 65504.f);
        amax = fmaxf(amax, fabsf(x[e]));
      }
      const h4 hi = __builtin_convertvector(x, h4);
      const h4 lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x4), h4);
      const u2 hu = __builtin_bit_cast(u2, hi), lu = __builtin_bit_cast(u2, lo);
      const bool odd = c4 & 1;
      // the even thread needs its partner's hi halves, the odd one its partner's lo halves
      const unsigned t0 = __shfl_xor(odd ? hu[0] : lu[0], 1), t1 = __shfl_xor(odd ? hu[1] : lu[1], 1);
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      const u4 piece = odd ? u4{t0, t1, lu[0], lu[1]} : u4{hu[0], hu[1], t0, t1};
      const unsigned urow = (unsigned)row, img = urow / hw, px = urow - img * hw;
      const int oct8 = c4 >> 1, cg = oct8 >> 1, oct = oct8 & 1, cg_total = c >> 4;      // (c % 16 == 0: no padded octet to zero)
      const size_t q = ((size_t)img * cg_total + cg) * 4 + (odd ? 2 : 0) + oct;
      *reinterpret_cast<u4*>(dz_sp + (q * hw + px) * 16) = piece;
    }
  }
  if constexpr (SP) {
    if (amax > 16384.f && flags) atomicOr(flags, amax >= 65504.f ? 3u : 2u);
  }
}

// The one-group fast form of bn_bwd_apply_v4_kernel (see bn_apply_v4_fast_kernel): per channel 1 / sqrtf(var + eps) and the two
// means (double sum / norm_rows, rounded to float -- eight fp64 divisions per float4 in the general kernel) once per workgroup
// into LDS, the same expressions; (row, channel quad) by shift and mask; the pixel decode of the incoming gradient in 32 bits.
// BIAS: the per-channel sums of dz -- the gradient of the conv bias in front of this BatchNorm -- leave this launch as one
// double per (workgroup, channel) in `bias_part` (folded by fold_partials_kernel in a fixed order: deterministic), instead of
// a dn_channel_sum pass that reads dz again (round 6: 0.5 ms of the det step).  A thread's channel quad never changes (the
// grid stride is a multiple of c / 4): it adds its own dz values in fp32 in loop order, the workgroup adds the threads of a
// quad in thread order in double.
template <bool SP, bool BIAS = false>
__global__ void __launch_bounds__(256)
bn_bwd_apply_v4_fast_kernel(GradSrc src, const float* __restrict__ z, const float* __restrict__ mean,
                            const float* __restrict__ var, const float* __restrict__ gamma, float eps, long norm_rows,
                            const double* __restrict__ sums, int sh, unsigned total4, float* __restrict__ dz,
                            unsigned char* __restrict__ dz_sp, float sp_lift, unsigned hw, unsigned* __restrict__ flags,
                            double* __restrict__ bias_part = nullptr) {
  __shared__ __attribute__((aligned(16))) float rstd_s[kMaxC], m1_s[kMaxC], m2_s[kMaxC];
  const int c = src.c;
  for (int i = threadIdx.x; i < c; i += 256) {
    rstd_s[i] = 1.f / sqrtf(var[i] + eps);
    m1_s[i] = (float)(sums[i] / norm_rows);
    m2_s[i] = (float)(sums[c + i] / norm_rows);
  }
  __syncthreads();
  const unsigned c4m = (1u << sh) - 1u;
  float amax = 0.f;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total4; idx += gridDim.x * 256u) {
    const unsigned row = idx >> sh;
    const int c4 = (int)(idx & c4m);
    const f32x4 rs = *reinterpret_cast<const f32x4*>(&rstd_s[4 * c4]);
    const f32x4 zh = (ldv4(z + (size_t)row * c + 4 * c4) - ldv4(mean + 4 * c4)) * rs;
    const f32x4 m1 = *reinterpret_cast<const f32x4*>(&m1_s[4 * c4]), m2 = *reinterpret_cast<const f32x4*>(&m2_s[4 * c4]);
    const f32x4 v = ldv4(gamma + 4 * c4) * rs * (src.v4u(row, c4) - m1 - zh * m2);
    if (!SP || dz) *reinterpret_cast<f32x4*>(dz + (size_t)row * c + 4 * c4) = v;      // (dz NULL: the SP copy is the only consumer's)
    if constexpr (BIAS) bsum += v;
    if constexpr (SP) {
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      typedef unsigned u2 __attribute__((ext_vector_type(2)));
      f32x4 x;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x[e] = fminf(fmaxf(v[e] * sp_lift, -65504.f), 65504.f);
        amax = fmaxf(amax, fabsf(x[e]));
      }
      const h4 hi = __builtin_convertvector(x, h4);
      const h4 lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x4), h4);
      const u2 hu = __builtin_bit_cast(u2, hi), lu = __builtin_bit_cast(u2, lo);
      const bool odd = c4 & 1;
      const unsigned t0 = __shfl_xor(odd ? hu[0] : lu[0], 1), t1 = __shfl_xor(odd ? hu[1] : lu[1], 1);
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      const u4 piece = odd ? u4{t0, t1, lu[0], lu[1]} : u4{hu[0], hu[1], t0, t1};
      const unsigned img = row / hw, px = row - img * hw;
      const int oct8 = c4 >> 1, cg = oct8 >> 1, oct = oct8 & 1, cg_total = c >> 4;
      const size_t q = ((size_t)img * cg_total + cg) * 4 + (odd ? 2 : 0) + oct;
      *reinterpret_cast<u4*>(dz_sp + (q * hw + px) * 16) = piece;
    }
  }
  if constexpr (SP) {
    if (amax > 16384.f && flags) atomicOr(flags, amax >= 65504.f ? 3u : 2u);
  }
  if constexpr (BIAS) {
    __shared__ __attribute__((aligned(16))) float bred[256][4];
    *reinterpret_cast<f32x4*>(bred[threadIdx.x]) = bsum;
    __syncthreads();
    const int c4n = 1 << sh;
    for (int cc = threadIdx.x; cc < c; cc += 256) {
      double t = 0.0;
      for (int k = cc >> 2; k < 256; k += c4n) t += (double)bred[k][cc & 3];      // the quad's threads, in thread order
      bias_part[(size_t)blockIdx.x * c + cc] = t;
    }
  }
}

__global__ void __launch_bounds__(256)
bn_bwd_reduce_kernel(GradSrc src, const float* __restrict__ z, const float* __restrict__ mean,
                     const float* __restrict__ var, float eps, long rows_per_group,
                     double* __restrict__ sums) {
  const int g = blockIdx.y, c = src.c;
  const long base = (long)g * rows_per_group;
  group_channel_sums<2>(c, rows_per_group, sums + (size_t)g * gridDim.x * 2 * c,
                        [&](long r, int cc, float& q0, float& q1) {
                          const float gr = src(base + r, cc);
                          const float rstd = 1.f / sqrtf(var[g * c + cc] + eps);
                          q0 = gr;
                          q1 = gr * ((z[(base + r) * c + cc] - mean[g * c + cc]) * rstd);
                        });
}

__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(GradSrc src, const float* __restrict__ z, const float* __restrict__ mean,
                    const float* __restrict__ var, const float* __restrict__ gamma, float eps,
                    long rows_per_group, long norm_rows, const double* __restrict__ sums, long total,
                    float* __restrict__ dz) {
  const int c = src.c;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / c;
    const int cc = (int)(idx % c);
    const int g = (int)(row / rows_per_group);
    const float rstd = 1.f / sqrtf(var[g * c + cc] + eps);
    const float zh = (z[idx] - mean[g * c + cc]) * rstd;
    const float m1 = (float)(sums[(size_t)g * 2 * c + cc] / norm_rows);
    const float m2 = (float)(sums[(size_t)g * 2 * c + c + cc] / norm_rows);
    dz[idx] = gamma[cc] * rstd * (src(row, cc) - m1 - zh * m2);
  }
}

__global__ void bn_param_grad_kernel(const double* __restrict__ sums, int n_groups, int c,
                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                     int accumulate) {
  const int cc = blockIdx.x * blockDim.x + threadIdx.x;
  if (cc >= c) return;
  double s1 = 0.0, s2 = 0.0;
  int g = 0;
  for (; g + 7 < n_groups; g += 8) {      // group order, eight groups' sums in flight
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = sums[(size_t)(g + u) * 2 * c + cc];
      b[u] = sums[(size_t)(g + u) * 2 * c + c + cc];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s1 += a[u];
      s2 += b[u];
    }
  }
  for (; g < n_groups; ++g) {
    s1 += sums[(size_t)g * 2 * c + cc];
    s2 += sums[(size_t)g * 2 * c + c + cc];
  }
  dbeta[cc] = (accumulate ? dbeta[cc] : 0.f) + (float)s1;
  dgamma[cc] = (accumulate ? dgamma[cc] : 0.f) + (float)s2;
}

__global__ void __launch_bounds__(256)
channel_sum_kernel(const float* __restrict__ x, long rows, int c, int ld, double* __restrict__ sums) {
  group_channel_sums<1>(c, rows, sums, [&](long r, int cc, float& q0, float& q1) {
    q0 = x[r * ld + cc];
    (void)q1;
  });
}

template <int U>
__global__ void __launch_bounds__(256)
channel_sum_v4_kernel(const float* __restrict__ x, long rows, int c, int ld, double* __restrict__ sums) {
  group_channel_sums_v4<1, U>(c, rows, sums, [&](long r, int c4, f32x4& q0, f32x4& q1) {
    q0 = ldv4(x + r * ld + 4 * c4);
    (void)q1;
  });
}

__global__ void channel_sum_finalize_kernel(const double* __restrict__ sums, int c,
                                            float* __restrict__ out, int accumulate) {
  const int cc = blockIdx.x * blockDim.x + threadIdx.x;
  if (cc < c) out[cc] = (accumulate ? out[cc] : 0.f) + (float)sums[cc];
}

__global__ void add_rows_kernel(float* __restrict__ a, int ld_a, const float* __restrict__ b, int ld_b,
                                int c, long total) {
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long row = idx / c;
    const int cc = (int)(idx % c);
    a[row * ld_a + cc] += b[row * ld_b + cc];
  }
}

// backward of F.interpolate(scale_factor=2, nearest) on its own: out[n, y, x, c] = sum of the 2 x 2
// block of the double-resolution gradient (a channel slice with row stride ld)
__global__ void upsample2_sum_kernel(const float* __restrict__ g, int ld, int h, int w, int c,
                                     long total, float* __restrict__ out) {
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(idx % c);
    long r = idx / c;
    const int px = (int)(r % w);
    r /= w;
    const int py = (int)(r % h);
    const long img = r / h;
    const float* b = g + ((img * 2 * h + 2 * py) * (2L * w) + 2 * px) * ld + cc;
    out[idx] = (b[0] + b[ld]) + (b[2L * w * ld] + b[(2L * w + 1) * ld]);
  }
}

// ---------------------------------------------------------------------------------
// fusion, training form
// ---------------------------------------------------------------------------------
__global__ void pair_add_ego_kernel(float* __restrict__ z1, const float* __restrict__ e,
                                    const int* __restrict__ ego_image, long per_image, long total) {
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int p = (int)(idx / per_image);
    z1[idx] += e[(size_t)ego_image[p] * per_image + idx % per_image];
  }
}

__global__ void pair_sum_ego_kernel(const float* __restrict__ dz1, const int* __restrict__ first,
                                    const int* __restrict__ pairs, long per_image, long total,
                                    float* __restrict__ de) {
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int img = (int)(idx / per_image);
    const long off = idx % per_image;
    float s = 0.f;
    for (int k = first[img]; k < first[img + 1]; ++k) s += dz1[(size_t)pairs[k] * per_image + off];
    de[idx] = s;
  }
}

constexpr int kMaxNbr = 16;

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one wavefront per (ego, pixel); lanes carry float4s of channels
__global__ void __launch_bounds__(256)
fuse_combine_kernel(const float* __restrict__ z4, const float* __restrict__ maps,
                    const int* __restrict__ first, const int* __restrict__ pair_index,
                    const int* __restrict__ map_image, const int* __restrict__ ego_out, int n_egos,
                    int hw, int c, float* __restrict__ weights, float* __restrict__ fused) {
  const int lane = threadIdx.x & 63;
  const long item = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (item >= (long)n_egos * hw) return;
  const int e = (int)(item / hw), px = (int)(item % hw);
  const int k0 = first[e];
  int kn = first[e + 1] - k0;
  kn = kn < 0 ? 0 : (kn > kMaxNbr ? kMaxNbr : kn);   // the per-lane arrays hold kMaxNbr entries (host side: train.py refuses more agents)
  float wk[kMaxNbr];
  float sum = 0.f;
  for (int k = 0; k < kn; ++k) {
    const int p = pair_index[k0 + k];
    const float s = p >= 0 ? fmaxf(z4[(size_t)p * hw + px], 0.f) : 0.f;
    wk[k] = expf(s);
    sum += wk[k];
  }
  for (int k = 0; k < kn; ++k) {
    wk[k] = wk[k] / sum;
    const int p = pair_index[k0 + k];
    if (lane == 0 && p >= 0) weights[(size_t)p * hw + px] = wk[k];
  }
  float* out = fused + ((size_t)ego_out[e] * hw + px) * c;
  for (int c4 = lane; c4 < (c >> 2); c4 += 64) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < kn; ++k)
      acc += *reinterpret_cast<const f32x4*>(maps + ((size_t)map_image[k0 + k] * hw + px) * c + 4 * c4) * wk[k];
    *reinterpret_cast<f32x4*>(out + 4 * c4) = acc;
  }
}

__global__ void __launch_bounds__(256)
fuse_combine_bwd_kernel(const float* __restrict__ dfused, int ld_df, const float* __restrict__ z4,
                        const float* __restrict__ weights, const float* __restrict__ maps,
                        const int* __restrict__ first, const int* __restrict__ pair_index,
                        const int* __restrict__ map_image, const int* __restrict__ ego_out, int n_egos,
                        int hw, int c, float* __restrict__ dmaps, float* __restrict__ dz4) {
  const int lane = threadIdx.x & 63;
  const long item = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (item >= (long)n_egos * hw) return;
  const int e = (int)(item / hw), px = (int)(item % hw);
  const int k0 = first[e];
  int kn = first[e + 1] - k0;
  kn = kn < 0 ? 0 : (kn > kMaxNbr ? kMaxNbr : kn);   // the per-lane arrays hold kMaxNbr entries (host side: train.py refuses more agents)
  const float* df = dfused + ((size_t)ego_out[e] * hw + px) * ld_df;
  float wk[kMaxNbr], dot[kMaxNbr];
  for (int k = 0; k < kn; ++k) {
    const int p = pair_index[k0 + k];
    wk[k] = p >= 0 ? weights[(size_t)p * hw + px] : 1.f;
    dot[k] = 0.f;
  }
  for (int c4 = lane; c4 < (c >> 2); c4 += 64) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(df + 4 * c4);
    for (int k = 0; k < kn; ++k) {
      const size_t off = ((size_t)map_image[k0 + k] * hw + px) * c + 4 * c4;
      const f32x4 m = *reinterpret_cast<const f32x4*>(maps + off);
      dot[k] += (g[0] * m[0] + g[1] * m[1]) + (g[2] * m[2] + g[3] * m[3]);
      *reinterpret_cast<f32x4*>(dmaps + off) = g * wk[k];
    }
  }
  float mean = 0.f;
  for (int k = 0; k < kn; ++k) {
    dot[k] = wave_sum(dot[k]);
    mean += wk[k] * dot[k];
  }
  if (lane == 0)
    for (int k = 0; k < kn; ++k) {
      const int p = pair_index[k0 + k];
      if (p >= 0) {
        const size_t i = (size_t)p * hw + px;
        dz4[i] = z4[i] > 0.f ? wk[k] * (dot[k] - mean) : 0.f;
      }
    }
}

// ---------------------------------------------------------------------------------
// loss and optimiser
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
det_loss_kernel(const float* __restrict__ cls, const float* __restrict__ labels,
                const float* __restrict__ loc, const float* __restrict__ targets,
                const float* __restrict__ mask, long n, int code, float alpha, float gamma,
                float sigma, float inv_norm, double* __restrict__ losses, float* __restrict__ dcls,
                float* __restrict__ dloc) {
  double l_cls = 0.0, l_loc = 0.0;
  const float s2 = sigma * sigma;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    // two-class softmax, target = the one-hot column
    const float z0 = cls[2 * i], z1 = cls[2 * i + 1];
    const float mx = fmaxf(z0, z1);
    const float e0 = expf(z0 - mx), e1 = expf(z1 - mx);
    const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
    const bool fg = labels[2 * i + 1] > 0.5f;
    const bool any = fg || labels[2 * i] > 0.5f;      // an all-zero row is "don't care"
    float d0 = 0.f, d1 = 0.f;
    if (any) {
      const float q = fg ? p1 : p0;
      const float a = fg ? alpha : 1.f - alpha;
      const float lq = logf(fmaxf(q, 1e-30f));
      const float om = 1.f - q;
      const float mod = powf(om, gamma);
      l_cls += (double)(-a * mod * lq);
      // dL/dq, then dq/dz_c = q (delta_ct - p_c)
      const float dmod = gamma == 0.f ? 0.f : gamma * powf(om, gamma - 1.f);
      const float dl_dq = a * (dmod * lq - mod / fmaxf(q, 1e-30f));
      const float t0 = fg ? 0.f : 1.f, t1 = fg ? 1.f : 0.f;
      d0 = dl_dq * q * (t0 - p0) * inv_norm;
      d1 = dl_dq * q * (t1 - p1) * inv_norm;
    }
    dcls[2 * i] = d0;
    dcls[2 * i + 1] = d1;
    const float mk = mask[i];
    for (int k = 0; k < code; ++k) {
      const float d = loc[i * code + k] - targets[i * code + k];
      const float ad = fabsf(d);
      float l, g;
      if (ad <= 1.f / s2) {
        l = 0.5f * s2 * d * d;
        g = s2 * d;
      } else {
        l = ad - 0.5f / s2;
        g = d > 0.f ? 1.f : -1.f;
      }
      l_loc += (double)(mk * l);
      dloc[i * code + k] = mk * g * inv_norm;
    }
  }
  // workgroup reduction, one atomic per workgroup and loss
  __shared__ double red[2][256];
  red[0][threadIdx.x] = l_cls;
  red[1][threadIdx.x] = l_loc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomic_add_f64(&losses[0], red[0][0] * inv_norm);
    atomic_add_f64(&losses[1], red[1][0] * inv_norm);
  }
}


// The same loss on FLAT float4 streams (round 6: the per-anchor form moved 786 MB per step at 54 % of the HBM rate -- every lane
// read its anchor's six box codes at a 24-byte stride).  Loop A: two anchors' class logits / labels per float4; loop B: four
// consecutive elements of loc / targets per float4, the anchor of element e is e / code (mask gather).  The per-element formulas
// are det_loss_kernel's, so dcls / dloc are bit for bit its values; the loss VALUES are summed in another order (they already
// leave through one f64 atomic per workgroup).  n even, code * n % 4 == 0, 16-byte aligned tensors.
__device__ inline void focal_one(float z0, float z1, float lb0, float lb1, float alpha, float gamma, float inv_norm,
                                 float& d0, float& d1, double& l_cls) {
  const float mx = fmaxf(z0, z1);
  const float e0 = expf(z0 - mx), e1 = expf(z1 - mx);
  const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
  const bool fg = lb1 > 0.5f;
  const bool any = fg || lb0 > 0.5f;
  d0 = 0.f; d1 = 0.f;
  if (any) {
    const float q = fg ? p1 : p0;
    const float a = fg ? alpha : 1.f - alpha;
    const float lq = logf(fmaxf(q, 1e-30f));
    const float om = 1.f - q;
    const float mod = powf(om, gamma);
    l_cls += (double)(-a * mod * lq);
    const float dmod = gamma == 0.f ? 0.f : gamma * powf(om, gamma - 1.f);
    const float dl_dq = a * (dmod * lq - mod / fmaxf(q, 1e-30f));
    const float t0 = fg ? 0.f : 1.f, t1 = fg ? 1.f : 0.f;
    d0 = dl_dq * q * (t0 - p0) * inv_norm;
    d1 = dl_dq * q * (t1 - p1) * inv_norm;
  }
}

__global__ void __launch_bounds__(256)
det_loss_v4_kernel(const float* __restrict__ cls, const float* __restrict__ labels, const float* __restrict__ loc,
                   const float* __restrict__ targets, const float* __restrict__ mask, long n, int code, float alpha, float gamma,
                   float sigma, float inv_norm, double* __restrict__ losses, float* __restrict__ dcls, float* __restrict__ dloc) {
  double l_cls = 0.0, l_loc = 0.0;
  const float s2 = sigma * sigma;
  const long stride = (long)gridDim.x * blockDim.x, t0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
  for (long i = t0; i < n / 2; i += stride) {
    const f32x4 z = ldv4(cls + 4 * i), lb = ldv4(labels + 4 * i);
    float d0, d1, d2, d3;
    focal_one(z[0], z[1], lb[0], lb[1], alpha, gamma, inv_norm, d0, d1, l_cls);
    focal_one(z[2], z[3], lb[2], lb[3], alpha, gamma, inv_norm, d2, d3, l_cls);
    *reinterpret_cast<f32x4*>(dcls + 4 * i) = f32x4{d0, d1, d2, d3};
  }
  const long m4 = n * code / 4;
  for (long i = t0; i < m4; i += stride) {
    const f32x4 x = ldv4(loc + 4 * i), tg = ldv4(targets + 4 * i);
    f32x4 g4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float mk = mask[(4 * i + e) / code];
      const float d = x[e] - tg[e];
      const float ad = fabsf(d);
      float l, g;
      if (ad <= 1.f / s2) {
        l = 0.5f * s2 * d * d;
        g = s2 * d;
      } else {
        l = ad - 0.5f / s2;
        g = d > 0.f ? 1.f : -1.f;
      }
      l_loc += (double)(mk * l);
      g4[e] = mk * g * inv_norm;
    }
    *reinterpret_cast<f32x4*>(dloc + 4 * i) = g4;
  }
  __shared__ double red[2][256];
  red[0][threadIdx.x] = l_cls;
  red[1][threadIdx.x] = l_loc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomic_add_f64(&losses[0], red[0][0] * inv_norm);
    atomic_add_f64(&losses[1], red[1][0] * inv_norm);
  }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                            float wd, float bc1, float bc2_sqrt) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i];
    if (wd != 0.f) gi += wd * p[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    // torch: denom = sqrt(v) / sqrt(bias_correction2) + eps; p -= (lr / bias_correction1) * m / denom
    p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
  }
}

// DN_BN_LEGACY=1: the general apply kernels and one row per iteration in the reductions (tests: the bitwise A/B; tools)
bool bn_legacy() {
  static const bool legacy = [] { const char* e = getenv("DN_BN_LEGACY"); return e && e[0] == '1'; }();
  return legacy;
}
// the one-group fast kernels (bn_apply_v4_fast_kernel, bn_bwd_apply_v4_fast_kernel): log2(c / 4) when they apply, else -1
int bn_fast_shift(int n_groups, int c, long total) {
  const int c4n = c >> 2;
  if (bn_legacy() || n_groups != 1 || c % 4 != 0 || c > kMaxC || c4n <= 0 || (c4n & (c4n - 1)) != 0 || total / 4 >= (1L << 31)) return -1;
  int sh = 0;
  while ((1 << sh) < c4n) ++sh;
  return sh;
}

int grid_for(long total, int cap = 4096) {
  const long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

bool vec4_ok(int c, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
  if (c % 4) return false;
  for (int ld : lds)
    if (ld % 4) return false;
  for (const void* p : ptrs)
    if (p && (reinterpret_cast<uintptr_t>(p) & 15)) return false;
  return true;
}

// Workgroups of a per-channel reduction over all groups.  1024 since round 5 (2048 before): measured in one lease
// (tools/r05_reduce_blocks.sh, profiles/r05_reduce_blocks.txt) the folds of a training step cost 0.69 / 0.48 / 0.36 ms at 2048 / 1024 /
// 512 and the reductions themselves 2.13 / 2.02 / 2.64 ms -- four workgroups per CU still saturate the HBM, two do not.
// DN_REDUCE_BLOCKS overrides (tools).
int reduce_blocks_total() {
  static const int n = [] {
    const char* e = getenv("DN_REDUCE_BLOCKS");
    const int v = e ? atoi(e) : 0;
    return v >= 64 && v <= 8192 ? v : 1024;
  }();
  return n;
}
int blocks_per_group(long rows_per_group, int n_groups) {
  // ~1024 workgroups over all groups, at least 64 rows each
  long b = reduce_blocks_total() / n_groups;
  if (b > rows_per_group / 64) b = rows_per_group / 64;
  return (int)(b < 1 ? 1 : b);
}

}  // namespace

// doubles of workspace a per-channel reduction needs: the folded sums + every workgroup's partial
extern "C" size_t dn_reduce_workspace_bytes(int n_groups, long rows_per_group, int c) {
  if (n_groups <= 0 || rows_per_group <= 0 || c <= 0) return 0;
  return sizeof(double) * 2 * c * n_groups * (size_t)(1 + blocks_per_group(rows_per_group, n_groups));
}

// workspace of the fused bias gradient of dn_bn_train_backward_finish_bias: [c] folded sums + one double per (workgroup of
// the apply launch, channel)
// workgroups of the apply launch when it also leaves the bias partials: every workgroup is one row of partials for the fold
// (one wavefront per channel walks them), so fewer than the plain launch's 8192 (DN_BN_BIAS_BLOCKS: tools)
static int bias_blocks_cap() {
  static const int cap = [] { const char* e = getenv("DN_BN_BIAS_BLOCKS"); const int v = e ? atoi(e) : 1024; return v < 64 ? 64 : (v > 8192 ? 8192 : v); }();
  return cap;
}
extern "C" size_t dn_bn_bias_workspace_bytes(long rows, int c) {
  if (rows <= 0 || c <= 0) return 0;
  return sizeof(double) * (size_t)c * (size_t)(1 + grid_for(rows * (long)c / 4, bias_blocks_cap()));
}

// Two-phase forms (round 5: agent-parallel training, sharded.py).  A rank that holds only SOME images of a BatchNorm batch
// reduces its own rows (`_partial`: the folded sums [n_groups][2 c] doubles land at the start of `sums`), the caller
// all-reduces those doubles over the ranks, and `_finish` normalises by the GLOBAL row count.  The one-call forms below are
// the two phases back to back with norm_rows = rows_per_group.
extern "C" int dn_bn_train_stats_partial(const float* z, int n_groups, long rows_per_group, int c, int ldz,
                                         double* sums, size_t sums_bytes, void* stream) {
  DN_REQUIRE(z && sums, "bn stats: null pointer");
  DN_REQUIRE(n_groups > 0 && rows_per_group > 0 && c > 0 && c <= kMaxC && ldz >= c,
             "bn stats: bad shape (groups %d rows %ld c %d ld %d)", n_groups, rows_per_group, c, ldz);
  DN_REQUIRE(sums_bytes >= dn_reduce_workspace_bytes(n_groups, rows_per_group, c),
             "bn stats: workspace of %zu bytes, dn_reduce_workspace_bytes() asks for %zu", sums_bytes,
             dn_reduce_workspace_bytes(n_groups, rows_per_group, c));
  hipStream_t s = (hipStream_t)stream;
  // workspace: [n_groups][2 c] folded sums, then the workgroups' partials [n_groups][blocks][2 c]
  const int nblk = blocks_per_group(rows_per_group, n_groups);
  double* part = sums + (size_t)2 * c * n_groups;
  if (vec4_ok(c, {ldz}, {z}))
    hipLaunchKernelGGL(bn_legacy() ? bn_stats_v4_kernel<1> : bn_stats_v4_kernel<4>, dim3(nblk, n_groups), dim3(256), 0, s, z, rows_per_group, c, ldz, part);
  else
    hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk, n_groups), dim3(256), 0, s, z, rows_per_group, c, ldz, part);
  hipLaunchKernelGGL(fold_partials_kernel, dim3((n_groups * 2 * c + 3) / 4), dim3(256), 0, s, part, nblk, 2 * c,
                     n_groups, sums);
  return dn::check_launch("bn_stats_kernel");
}

extern "C" int dn_bn_train_stats_finish(const double* sums, int n_groups, long norm_rows, int c, float* mean, float* var,
                                        void* stream) {
  DN_REQUIRE(sums && mean && var, "bn stats finish: null pointer");
  DN_REQUIRE(n_groups > 0 && norm_rows > 0 && c > 0 && c <= kMaxC, "bn stats finish: bad shape");
  const int n = n_groups * c;
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, n, c,
                     norm_rows, mean, var);
  return dn::check_launch("bn_stats_finalize_kernel");
}

namespace {
int bn_stats_one_call(const float* z, int n_groups, long rows_per_group, int c, int ldz, double* sums, size_t sums_bytes,
                      float* mean, float* var, float* rmean, float* rvar, float momentum, long unbias_rows, void* stream) {
  DN_REQUIRE(z && sums && mean && var, "bn stats: null pointer");
  DN_REQUIRE(n_groups > 0 && rows_per_group > 0 && c > 0 && c <= kMaxC && ldz >= c, "bn stats: bad shape");
  DN_REQUIRE(sums_bytes >= dn_reduce_workspace_bytes(n_groups, rows_per_group, c),
             "bn stats: workspace of %zu bytes, dn_reduce_workspace_bytes() asks for %zu", sums_bytes,
             dn_reduce_workspace_bytes(n_groups, rows_per_group, c));
  DN_REQUIRE(!rmean || (rvar && n_groups == 1), "bn stats: the fused running-statistics update takes one group");
  hipStream_t s = (hipStream_t)stream;
  const int nblk = blocks_per_group(rows_per_group, n_groups);
  double* part = sums + (size_t)2 * c * n_groups;      // workspace layout: see dn_bn_train_stats_partial
  if (vec4_ok(c, {ldz}, {z}))
    hipLaunchKernelGGL(bn_legacy() ? bn_stats_v4_kernel<1> : bn_stats_v4_kernel<4>, dim3(nblk, n_groups), dim3(256), 0, s, z, rows_per_group, c, ldz, part);
  else
    hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk, n_groups), dim3(256), 0, s, z, rows_per_group, c, ldz, part);
  // fold + finish (+ the running statistics) in one launch: the sums and statistics of the two-launch path, bit for bit
  hipLaunchKernelGGL(fold_stats_finish_kernel, dim3((n_groups * c + 3) / 4), dim3(256), 0, s, part, nblk, c, n_groups, rows_per_group, sums,
                     mean, var, rmean, rvar, momentum, unbias_rows);
  return dn::check_launch("bn_stats_kernel");
}
}  // namespace

extern "C" int dn_bn_train_stats(const float* z, int n_groups, long rows_per_group, int c, int ldz,
                                 double* sums, size_t sums_bytes, float* mean, float* var, void* stream) {
  return bn_stats_one_call(z, n_groups, rows_per_group, c, ldz, sums, sums_bytes, mean, var, nullptr, nullptr, 0.f, 0, stream);
}

extern "C" int dn_bn_train_stats_running(const float* z, long rows, int c, int ldz, double* sums, size_t sums_bytes, float* mean,
                                         float* var, float* running_mean, float* running_var, float momentum, void* stream) {
  DN_REQUIRE(running_mean && running_var, "bn stats (+ running): null pointer");
  return bn_stats_one_call(z, 1, rows, c, ldz, sums, sums_bytes, mean, var, running_mean, running_var, momentum, rows, stream);
}

extern "C" int dn_bn_train_apply(const float* z, const float* mean, const float* var,
                                 const float* gamma, const float* beta, float eps, int relu,
                                 int n_groups, long rows_per_group, int c, int ldz, float* y,
                                 void* stream) {
  DN_REQUIRE(z && mean && var && gamma && beta && y, "bn apply: null pointer");
  DN_REQUIRE(n_groups > 0 && rows_per_group > 0 && c > 0 && ldz >= c, "bn apply: bad shape");
  const long total = (long)n_groups * rows_per_group * c;
  const int fsh = bn_fast_shift(n_groups, c, total);
  if (vec4_ok(c, {ldz}, {z, y, mean, var, gamma, beta}) && fsh >= 0)
    hipLaunchKernelGGL(bn_apply_v4_fast_kernel<false>, dim3(grid_for(total / 4, 8192)), dim3(256), 0, (hipStream_t)stream, z, mean, var,
                       gamma, beta, eps, relu, c, fsh, ldz, (unsigned)(total / 4), y, (unsigned char*)nullptr,
                       (unsigned char*)nullptr, 1u, (unsigned*)nullptr);
  else if (vec4_ok(c, {ldz}, {z, y, mean, var, gamma, beta}))
    hipLaunchKernelGGL(bn_apply_v4_kernel, dim3(grid_for(total / 4, 8192)), dim3(256), 0,
                       (hipStream_t)stream, z, mean, var, gamma, beta, eps, relu, rows_per_group, c, ldz,
                       total / 4, y, (unsigned char*)nullptr);
  else
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream, z,
                       mean, var, gamma, beta, eps, relu, rows_per_group, c, ldz, total, y);
  return dn::check_launch("bn_apply_kernel");
}

extern "C" int dn_bn_train_apply_mask(const float* z, const float* mean, const float* var, const float* gamma,
                                      const float* beta, float eps, int n_groups, long rows_per_group, int c, int ldz,
                                      float* y, unsigned char* relu_mask, void* stream) {
  DN_REQUIRE(z && mean && var && gamma && beta && y && relu_mask, "bn apply (mask): null pointer");
  DN_REQUIRE(n_groups > 0 && rows_per_group > 0 && c > 0 && ldz >= c, "bn apply (mask): bad shape");
  DN_REQUIRE(vec4_ok(c, {ldz}, {z, y, mean, var, gamma, beta}), "bn apply (mask): needs c %% 4 == 0 and 16-byte aligned tensors");
  const long total = (long)n_groups * rows_per_group * c;
  const int fsh = bn_fast_shift(n_groups, c, total);
  if (fsh >= 0)
    hipLaunchKernelGGL(bn_apply_v4_fast_kernel<false>, dim3(grid_for(total / 4, 8192)), dim3(256), 0, (hipStream_t)stream, z, mean, var,
                       gamma, beta, eps, 1, c, fsh, ldz, (unsigned)(total / 4), y, relu_mask, (unsigned char*)nullptr, 1u,
                       (unsigned*)nullptr);
  else
    hipLaunchKernelGGL(bn_apply_v4_kernel, dim3(grid_for(total / 4, 8192)), dim3(256), 0, (hipStream_t)stream, z, mean, var,
                       gamma, beta, eps, 1, rows_per_group, c, ldz, total / 4, y, relu_mask);
  return dn::check_launch("bn_apply_kernel (mask)");
}

extern "C" int dn_bn_train_apply_mask_sp(const float* z, const float* mean, const float* var, const float* gamma,
                                         const float* beta, float eps, long rows, int hw, int c, int ldz, float* y,
                                         unsigned char* relu_mask, void* y_sp, void* stream) {
  DN_REQUIRE(z && mean && var && gamma && beta && y && relu_mask && y_sp, "bn apply (mask + SP): null pointer");
  DN_REQUIRE(rows > 0 && hw > 0 && rows % hw == 0 && c > 0 && c % 16 == 0 && ldz >= c, "bn apply (mask + SP): bad shape");
  DN_REQUIRE(vec4_ok(c, {ldz}, {z, y, mean, var, gamma, beta}) && (reinterpret_cast<uintptr_t>(y_sp) & 15) == 0,
             "bn apply (mask + SP): needs 16-byte aligned tensors");
  const long total = rows * c;
  const int c4n = c >> 2;      // (its own test: the SP form has no general-kernel twin for DN_BN_LEGACY to select)
  DN_REQUIRE(c <= kMaxC && (c4n & (c4n - 1)) == 0 && total / 4 < (1L << 31),
             "bn apply (mask + SP): c / 4 must be a power of two and the map below 2^31 float4s (c = %d)", c);
  int fsh = 0;
  while ((1 << fsh) < c4n) ++fsh;
  unsigned* flags = dn::sp_range_word();
  DN_REQUIRE(flags, "bn apply (mask + SP): the range word of the split-f16 engine is not addressable");
  hipLaunchKernelGGL(bn_apply_v4_fast_kernel<true>, dim3(grid_for(total / 4, 8192)), dim3(256), 0, (hipStream_t)stream, z, mean, var,
                     gamma, beta, eps, 1, c, fsh, ldz, (unsigned)(total / 4), y, relu_mask, (unsigned char*)y_sp, (unsigned)hw, flags);
  return dn::check_launch("bn_apply_kernel (mask + SP)");
}

extern "C" int dn_bn_update_running(const float* mean, const float* var, int n_groups,
                                    long rows_per_group, int c, const int* order, float momentum,
                                    float* running_mean, float* running_var, void* stream) {
  DN_REQUIRE(mean && var && running_mean && running_var, "bn running: null pointer");
  DN_REQUIRE(n_groups > 0 && rows_per_group > 0 && c > 0, "bn running: bad shape");
  hipLaunchKernelGGL(bn_update_running_kernel, dim3((c + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                     mean, var, n_groups, rows_per_group, c, order, momentum, running_mean, running_var);
  return dn::check_launch("bn_update_running_kernel");
}

// phase 1 of the backward: this rank's sums of g and g * zhat (folded, at the start of `sums`) and the parameter gradients
// out of THESE rows (dgamma / dbeta are sums over rows: ranks add theirs with the gradient all-reduce)
extern "C" int dn_bn_train_backward_partial(const float* dy_a, int ld_a, int up_a, const float* dy_b, int ld_b,
                                            const float* y, const float* z, const float* mean, const float* var, float eps,
                                            int relu, int n_groups, int h, int w, int images_per_group, int c, double* sums,
                                            size_t sums_bytes, float* dgamma, float* dbeta, int accumulate, void* stream) {
  DN_REQUIRE(dy_a && z && mean && var && sums && dgamma && dbeta, "bn backward: null pointer");
  DN_REQUIRE(!relu || y, "bn backward: relu needs y");
  DN_REQUIRE(relu >= 0 && relu <= 2 && (relu != 2 || c % 4 == 0), "bn backward: relu = 2 (y is the byte mask) needs c %% 4 == 0");
  DN_REQUIRE(up_a >= 0 && up_a <= 2 && (up_a != 2 || (h % 2 == 0 && w % 2 == 0 && ld_a >= 4 * c)),
             "bn backward: up_a = 2 (dy_a is the space-to-depth image [h / 2][w / 2][4 c]) needs even h, w and ld_a >= 4 c");
  DN_REQUIRE(n_groups > 0 && h > 0 && w > 0 && images_per_group > 0 && c > 0 && c <= kMaxC &&
                 ld_a >= c && (!dy_b || ld_b >= c),
             "bn backward: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const long rows_per_group = (long)images_per_group * h * w;
  DN_REQUIRE(sums_bytes >= dn_reduce_workspace_bytes(n_groups, rows_per_group, c),
             "bn backward: workspace of %zu bytes, dn_reduce_workspace_bytes() asks for %zu", sums_bytes,
             dn_reduce_workspace_bytes(n_groups, rows_per_group, c));
  GradSrc src{dy_a, dy_b, y, ld_a, up_a, ld_b, relu, h, w, c};
  const int nblk = blocks_per_group(rows_per_group, n_groups);
  double* part = sums + (size_t)2 * c * n_groups;      // workspace layout: see dn_bn_train_stats_partial
  if (vec4_ok(c, {ld_a, dy_b ? ld_b : 0}, {dy_a, dy_b, y, z, mean, var}))
    hipLaunchKernelGGL(bn_legacy() ? bn_bwd_reduce_v4_kernel<1> : bn_bwd_reduce_v4_kernel<2>, dim3(nblk, n_groups), dim3(256), 0, s, src, z, mean, var, eps,
                       rows_per_group, part);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk, n_groups), dim3(256), 0, s, src, z, mean, var, eps,
                       rows_per_group, part);
  if (n_groups == 1 && !bn_legacy()) {      // fold + parameter gradients in one launch (same sums, same bits)
    hipLaunchKernelGGL(fold_param_grad_kernel, dim3((c + 3) / 4), dim3(256), 0, s, part, nblk, c, sums, dgamma, dbeta, accumulate);
    return dn::check_launch("bn_backward reduce kernels");
  }
  hipLaunchKernelGGL(fold_partials_kernel, dim3((n_groups * 2 * c + 3) / 4), dim3(256), 0, s, part, nblk, 2 * c,
                     n_groups, sums);
  hipLaunchKernelGGL(bn_param_grad_kernel, dim3((c + 63) / 64), dim3(64), 0, s, sums, n_groups, c,
                     dgamma, dbeta, accumulate);
  return dn::check_launch("bn_backward reduce kernels");
}

// phase 2: dz of this rank's rows from the (all-reduced) sums, means taken over norm_rows rows per group
namespace {
int bn_backward_finish_impl(const float* dy_a, int ld_a, int up_a, const float* dy_b, int ld_b,
                            const float* y, const float* z, const float* mean, const float* var,
                            const float* gamma, float eps, int relu, int n_groups, int h, int w,
                            int images_per_group, int c, const double* sums, long norm_rows, float* dz,
                            void* dz_sp, float sp_lift, void* stream, float* dbias = nullptr, double* bias_ws = nullptr,
                            size_t bias_ws_bytes = 0, int* defer_blocks = nullptr) {
  DN_REQUIRE(dy_a && z && mean && var && gamma && sums && (dz || dz_sp), "bn backward finish: null pointer");
  DN_REQUIRE(!relu || y, "bn backward: relu needs y");
  // dz NULL (round 6): only the SP copy is written -- for a layer whose weight gradient (dn_conv_wgrad_sp_z), data gradient and
  // bias gradient all read dz through this launch's other outputs; the one-group fast kernels only
  DN_REQUIRE(dz || (n_groups == 1 && bn_fast_shift(n_groups, c, (long)images_per_group * h * w * c) >= 0),
             "bn backward: dz may be NULL only where the one-group fast form runs (one group, c / 4 a power of two; DN_BN_LEGACY unset)");
  DN_REQUIRE(relu >= 0 && relu <= 2 && (relu != 2 || c % 4 == 0), "bn backward: relu = 2 (y is the byte mask) needs c %% 4 == 0");
  DN_REQUIRE(up_a >= 0 && up_a <= 2 && (up_a != 2 || (h % 2 == 0 && w % 2 == 0 && ld_a >= 4 * c)),
             "bn backward: up_a = 2 (dy_a is the space-to-depth image [h / 2][w / 2][4 c]) needs even h, w and ld_a >= 4 c");
  DN_REQUIRE(n_groups > 0 && h > 0 && w > 0 && images_per_group > 0 && c > 0 && c <= kMaxC && norm_rows > 0 &&
                 ld_a >= c && (!dy_b || ld_b >= c),
             "bn backward finish: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const long rows_per_group = (long)images_per_group * h * w;
  GradSrc src{dy_a, dy_b, y, ld_a, up_a, ld_b, relu, h, w, c};
  const long total = (long)n_groups * rows_per_group * c;
  // the fused bias gradient (round 6): the apply launch leaves one double per (workgroup, channel), folded in a fixed order
  auto bias_launch = [&](auto sp_c, void* sp_ptr, float lift, unsigned* fl, int fsh) -> int {
    constexpr bool SPF = decltype(sp_c)::value;
    const int blocks = grid_for(total / 4, bias_blocks_cap());
    DN_REQUIRE(bias_ws && bias_ws_bytes >= dn_bn_bias_workspace_bytes(rows_per_group, c),
               "bn backward: bias workspace of %zu bytes, dn_bn_bias_workspace_bytes() asks for %zu", bias_ws_bytes,
               dn_bn_bias_workspace_bytes(rows_per_group, c));
    double* part = bias_ws + c;      // [c] folded sums, then [blocks][c] partials
    hipLaunchKernelGGL((bn_bwd_apply_v4_fast_kernel<SPF, true>), dim3(blocks), dim3(256), 0, s, src, z, mean, var, gamma, eps,
                       norm_rows, sums, fsh, (unsigned)(total / 4), dz, (unsigned char*)sp_ptr, lift, (unsigned)(h * w), fl, part);
    if (defer_blocks)      // the fold is the caller's (dn_channel_sum_fold_multi): the partials stay in bias_ws
      *defer_blocks = blocks;
    else
      hipLaunchKernelGGL(fold_channel_sum_kernel, dim3((c + 3) / 4), dim3(256), 0, s, part, blocks, c, bias_ws, dbias, 0);
    return dn::check_launch("bn backward apply kernel (+ bias gradient)");
  };
  if (dz_sp) {
    DN_REQUIRE(n_groups == 1 && c % 16 == 0 && vec4_ok(c, {ld_a, dy_b ? ld_b : 0}, {dy_a, dy_b, y, z, mean, var, gamma, dz}) &&
                   (reinterpret_cast<uintptr_t>(dz_sp) & 15) == 0 && rows_per_group < (1L << 31),
               "bn backward: the SP copy of dz needs one group, c %% 16 == 0, 16-byte aligned tensors (c = %d, groups = %d)", c, n_groups);
    DN_REQUIRE(sp_lift > 0.f && std::isfinite(sp_lift), "bn backward: sp_lift must be a positive finite power of two");
    unsigned* flags = dn::sp_range_word();
    DN_REQUIRE(flags, "bn backward: the range word of the split-f16 engine is not addressable");
    const int fsh = bn_fast_shift(n_groups, c, total);
    if (fsh >= 0 && (dbias || defer_blocks))
      return bias_launch(std::true_type{}, dz_sp, sp_lift, flags, fsh);
    DN_REQUIRE(!dbias && !defer_blocks, "bn backward: the fused bias gradient needs the one-group fast form (c / 4 a power of two; DN_BN_LEGACY unset)");
    if (fsh >= 0)
      hipLaunchKernelGGL(bn_bwd_apply_v4_fast_kernel<true>, dim3(grid_for(total / 4, 8192)), dim3(256), 0, s, src, z, mean, var, gamma,
                         eps, norm_rows, sums, fsh, (unsigned)(total / 4), dz, (unsigned char*)dz_sp, sp_lift, (unsigned)(h * w), flags);
    else
      hipLaunchKernelGGL(bn_bwd_apply_v4_kernel<true>, dim3(grid_for(total / 4, 8192)), dim3(256), 0, s, src, z, mean, var, gamma,
                         eps, rows_per_group, norm_rows, sums, total / 4, dz, (unsigned char*)dz_sp, sp_lift, (unsigned)(h * w), flags);
    return dn::check_launch("bn backward apply kernel (SP copy)");
  }
  const int fsh = bn_fast_shift(n_groups, c, total);
  if (dbias || defer_blocks) {
    DN_REQUIRE(vec4_ok(c, {ld_a, dy_b ? ld_b : 0}, {dy_a, dy_b, y, z, mean, var, gamma, dz}) && fsh >= 0,
               "bn backward: the fused bias gradient needs the one-group fast form (c / 4 a power of two, aligned tensors; DN_BN_LEGACY unset)");
    return bias_launch(std::false_type{}, nullptr, 1.f, nullptr, fsh);
  }
  if (vec4_ok(c, {ld_a, dy_b ? ld_b : 0}, {dy_a, dy_b, y, z, mean, var, gamma, dz}) && fsh >= 0)
    hipLaunchKernelGGL(bn_bwd_apply_v4_fast_kernel<false>, dim3(grid_for(total / 4, 8192)), dim3(256), 0, s, src, z, mean, var, gamma,
                       eps, norm_rows, sums, fsh, (unsigned)(total / 4), dz, nullptr, 1.f, 1u, nullptr);
  else if (vec4_ok(c, {ld_a, dy_b ? ld_b : 0}, {dy_a, dy_b, y, z, mean, var, gamma, dz}))
    hipLaunchKernelGGL(bn_bwd_apply_v4_kernel<false>, dim3(grid_for(total / 4, 8192)), dim3(256), 0, s, src, z,
                       mean, var, gamma, eps, rows_per_group, norm_rows, sums, total / 4, dz, nullptr, 1.f, 1u, nullptr);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, s, src, z, mean,
                       var, gamma, eps, rows_per_group, norm_rows, sums, total, dz);
  return dn::check_launch("bn_backward apply kernels");
}
}  // namespace

extern "C" int dn_bn_train_backward_finish_bias(const float* dy_a, int ld_a, int up_a, const float* dy_b, int ld_b,
                                                const float* y, const float* z, const float* mean, const float* var,
                                                const float* gamma, float eps, int relu, int h, int w, int images, int c,
                                                const double* sums, long norm_rows, float* dz, void* dz_sp, float sp_lift,
                                                float* dbias, double* bias_ws, size_t bias_ws_bytes, void* stream) {
  DN_REQUIRE(dbias && bias_ws, "bn backward finish (+ bias gradient): null pointer");
  return bn_backward_finish_impl(dy_a, ld_a, up_a, dy_b, ld_b, y, z, mean, var, gamma, eps, relu, 1, h, w, images, c, sums,
                                 norm_rows, dz, dz_sp, sp_lift, stream, dbias, bias_ws, bias_ws_bytes);
}

extern "C" int dn_bn_train_backward_finish_bias_deferred(const float* dy_a, int ld_a, int up_a, const float* dy_b, int ld_b,
                                                         const float* y, const float* z, const float* mean, const float* var,
                                                         const float* gamma, float eps, int relu, int h, int w, int images, int c,
                                                         const double* sums, long norm_rows, float* dz, void* dz_sp, float sp_lift,
                                                         double* bias_ws, size_t bias_ws_bytes, int* n_blocks, void* stream) {
  DN_REQUIRE(bias_ws && n_blocks, "bn backward finish (bias gradient deferred): null pointer");
  return bn_backward_finish_impl(dy_a, ld_a, up_a, dy_b, ld_b, y, z, mean, var, gamma, eps, relu, 1, h, w, images, c, sums,
                                 norm_rows, dz, dz_sp, sp_lift, stream, nullptr, bias_ws, bias_ws_bytes, n_blocks);
}

extern "C" int dn_channel_sum_fold_multi(const dn_fold_job* jobs, int n_jobs, void* stream) {
  DN_REQUIRE(jobs && n_jobs > 0, "channel sum fold multi: no jobs");
  for (int j0 = 0; j0 < n_jobs; j0 += kFoldJobs) {
    const int n = n_jobs - j0 < kFoldJobs ? n_jobs - j0 : kFoldJobs;
    FoldJobs J;
    int waves = 0;
    for (int j = 0; j < n; ++j) {
      const dn_fold_job& q = jobs[j0 + j];
      DN_REQUIRE(q.partials && q.sums && q.out && q.n_blocks > 0 && q.c > 0 && q.c <= kMaxC, "channel sum fold multi: job %d", j0 + j);
      J.part[j] = q.partials; J.sums[j] = q.sums; J.out[j] = q.out;
      J.n_blocks[j] = q.n_blocks; J.c[j] = q.c; J.accumulate[j] = q.accumulate;
      J.first[j] = waves;
      waves += q.c;
    }
    for (int j = n; j <= kFoldJobs; ++j) J.first[j] = waves;
    for (int j = n; j < kFoldJobs; ++j) { J.part[j] = nullptr; J.sums[j] = nullptr; J.out[j] = nullptr; J.n_blocks[j] = J.c[j] = J.accumulate[j] = 0; }
    hipLaunchKernelGGL(fold_channel_sum_multi_kernel, dim3((waves + 3) / 4), dim3(256), 0, (hipStream_t)stream, J, n);
  }
  return dn::check_launch("fold_channel_sum_multi_kernel");
}

extern "C" int dn_bn_train_backward_finish(const float* dy_a, int ld_a, int up_a, const float* dy_b, int ld_b,
                                           const float* y, const float* z, const float* mean, const float* var,
                                           const float* gamma, float eps, int relu, int n_groups, int h, int w,
                                           int images_per_group, int c, const double* sums, long norm_rows, float* dz,
                                           void* stream) {
  return bn_backward_finish_impl(dy_a, ld_a, up_a, dy_b, ld_b, y, z, mean, var, gamma, eps, relu, n_groups, h, w,
                                 images_per_group, c, sums, norm_rows, dz, nullptr, 1.f, stream);
}

extern "C" int dn_bn_train_backward_finish_sp(const float* dy_a, int ld_a, int up_a, const float* dy_b, int ld_b,
                                              const float* y, const float* z, const float* mean, const float* var,
                                              const float* gamma, float eps, int relu, int n_groups, int h, int w,
                                              int images_per_group, int c, const double* sums, long norm_rows, float* dz,
                                              void* dz_sp, float sp_lift, void* stream) {
  DN_REQUIRE(dz_sp, "bn backward finish (SP copy): null pointer");
  return bn_backward_finish_impl(dy_a, ld_a, up_a, dy_b, ld_b, y, z, mean, var, gamma, eps, relu, n_groups, h, w,
                                 images_per_group, c, sums, norm_rows, dz, dz_sp, sp_lift, stream);
}

extern "C" int dn_bn_train_backward(const float* dy_a, int ld_a, int up_a, const float* dy_b, int ld_b,
                                    const float* y, const float* z, const float* mean,
                                    const float* var, const float* gamma, float eps, int relu,
                                    int n_groups, int h, int w, int images_per_group, int c,
                                    double* sums, size_t sums_bytes, float* dz, float* dgamma, float* dbeta,
                                    int accumulate, void* stream) {
  DN_REQUIRE(gamma && dz, "bn backward: null pointer");
  if (int rc = dn_bn_train_backward_partial(dy_a, ld_a, up_a, dy_b, ld_b, y, z, mean, var, eps, relu, n_groups, h, w,
                                            images_per_group, c, sums, sums_bytes, dgamma, dbeta, accumulate, stream))
    return rc;
  return dn_bn_train_backward_finish(dy_a, ld_a, up_a, dy_b, ld_b, y, z, mean, var, gamma, eps, relu, n_groups, h, w,
                                     images_per_group, c, sums, (long)images_per_group * h * w, dz, stream);
}

extern "C" int dn_channel_sum(const float* x, long rows, int c, int ld, double* sums, size_t sums_bytes, float* out,
                              int accumulate, void* stream) {
  DN_REQUIRE(x && sums && out, "channel sum: null pointer");
  DN_REQUIRE(rows > 0 && c > 0 && c <= kMaxC && ld >= c, "channel sum: bad shape");
  DN_REQUIRE(sums_bytes >= dn_reduce_workspace_bytes(1, rows, c),
             "channel sum: workspace of %zu bytes, dn_reduce_workspace_bytes() asks for %zu", sums_bytes,
             dn_reduce_workspace_bytes(1, rows, c));
  hipStream_t s = (hipStream_t)stream;
  const int nblk = blocks_per_group(rows, 1);
  double* part = sums + c;                              // [c] folded sums, then the workgroups' partials [blocks][c]
  if (vec4_ok(c, {ld}, {x}))
    hipLaunchKernelGGL(bn_legacy() ? channel_sum_v4_kernel<1> : channel_sum_v4_kernel<4>, dim3(nblk), dim3(256), 0, s, x, rows, c, ld, part);
  else
    hipLaunchKernelGGL(channel_sum_kernel, dim3(nblk), dim3(256), 0, s, x, rows, c, ld, part);
  hipLaunchKernelGGL(fold_channel_sum_kernel, dim3((c + 3) / 4), dim3(256), 0, s, part, nblk, c, sums, out, accumulate);
  return dn::check_launch("channel_sum_kernel");
}

extern "C" int dn_channel_sum_partial(const float* x, long rows, int c, int ld, double* sums, size_t sums_bytes, int* n_blocks,
                                      void* stream) {
  DN_REQUIRE(x && sums && n_blocks, "channel sum (partial): null pointer");
  DN_REQUIRE(rows > 0 && c > 0 && c <= kMaxC && ld >= c, "channel sum: bad shape");
  DN_REQUIRE(sums_bytes >= dn_reduce_workspace_bytes(1, rows, c),
             "channel sum: workspace of %zu bytes, dn_reduce_workspace_bytes() asks for %zu", sums_bytes,
             dn_reduce_workspace_bytes(1, rows, c));
  hipStream_t s = (hipStream_t)stream;
  const int nblk = blocks_per_group(rows, 1);
  double* part = sums + c;                              // [c] folded sums (the fold's), then the workgroups' partials [blocks][c]
  if (vec4_ok(c, {ld}, {x}))
    hipLaunchKernelGGL(bn_legacy() ? channel_sum_v4_kernel<1> : channel_sum_v4_kernel<4>, dim3(nblk), dim3(256), 0, s, x, rows, c, ld, part);
  else
    hipLaunchKernelGGL(channel_sum_kernel, dim3(nblk), dim3(256), 0, s, x, rows, c, ld, part);
  *n_blocks = nblk;
  return dn::check_launch("channel_sum_kernel");
}

extern "C" int dn_add_rows(float* a, int ld_a, const float* b, int ld_b, long rows, int c,
                           void* stream) {
  DN_REQUIRE(a && b && rows > 0 && c > 0 && ld_a >= c && ld_b >= c, "add rows: bad arguments");
  const long total = rows * c;
  hipLaunchKernelGGL(add_rows_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream, a,
                     ld_a, b, ld_b, c, total);
  return dn::check_launch("add_rows_kernel");
}

extern "C" int dn_upsample2_sum(const float* g, int ld, int n_images, int h, int w, int c, float* out,
                                void* stream) {
  DN_REQUIRE(g && out && n_images > 0 && h > 0 && w > 0 && c > 0 && ld >= c, "upsample sum: bad arguments");
  const long total = (long)n_images * h * w * c;
  hipLaunchKernelGGL(upsample2_sum_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream,
                     g, ld, h, w, c, total, out);
  return dn::check_launch("upsample2_sum_kernel");
}

extern "C" int dn_pair_add_ego(float* z1, const float* e, const int* ego_image, int n_pairs,
                               int rows_per_image, int c, void* stream) {
  DN_REQUIRE(z1 && e && ego_image && n_pairs > 0 && rows_per_image > 0 && c > 0,
             "pair add: bad arguments");
  const long per_image = (long)rows_per_image * c, total = per_image * n_pairs;
  hipLaunchKernelGGL(pair_add_ego_kernel, dim3(grid_for(total, 8192)), dim3(256), 0,
                     (hipStream_t)stream, z1, e, ego_image, per_image, total);
  return dn::check_launch("pair_add_ego_kernel");
}

extern "C" int dn_pair_sum_ego(const float* dz1, const int* first, const int* pairs, int n_images,
                               int rows_per_image, int c, float* de, void* stream) {
  DN_REQUIRE(dz1 && first && pairs && de && n_images > 0 && rows_per_image > 0 && c > 0,
             "pair sum: bad arguments");
  const long per_image = (long)rows_per_image * c, total = per_image * n_images;
  hipLaunchKernelGGL(pair_sum_ego_kernel, dim3(grid_for(total, 8192)), dim3(256), 0,
                     (hipStream_t)stream, dz1, first, pairs, per_image, total, de);
  return dn::check_launch("pair_sum_ego_kernel");
}

extern "C" int dn_fuse_combine(const float* z4, const float* maps, const int* first,
                               const int* pair_index, const int* map_image, const int* ego_out,
                               int n_egos, int hw, int c, float* weights, float* fused, void* stream) {
  DN_REQUIRE(z4 && maps && first && pair_index && map_image && ego_out && weights && fused,
             "fuse combine: null pointer");
  DN_REQUIRE(n_egos > 0 && hw > 0 && c > 0 && c % 4 == 0, "fuse combine: bad shape");
  const long items = (long)n_egos * hw;
  hipLaunchKernelGGL(fuse_combine_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, z4, maps, first, pair_index, map_image, ego_out, n_egos, hw, c,
                     weights, fused);
  return dn::check_launch("fuse_combine_kernel");
}

extern "C" int dn_fuse_combine_backward(const float* dfused, int ld_df, const float* z4,
                                        const float* weights, const float* maps, const int* first,
                                        const int* pair_index, const int* map_image,
                                        const int* ego_out, int n_egos, int hw, int c, float* dmaps,
                                        float* dz4, void* stream) {
  DN_REQUIRE(dfused && z4 && weights && maps && first && pair_index && map_image && ego_out && dmaps &&
                 dz4,
             "fuse combine backward: null pointer");
  DN_REQUIRE(n_egos > 0 && hw > 0 && c > 0 && c % 4 == 0 && ld_df >= c && ld_df % 4 == 0,
             "fuse combine backward: bad shape");
  const long items = (long)n_egos * hw;
  hipLaunchKernelGGL(fuse_combine_bwd_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, dfused, ld_df, z4, weights, maps, first, pair_index, map_image,
                     ego_out, n_egos, hw, c, dmaps, dz4);
  return dn::check_launch("fuse_combine_bwd_kernel");
}

extern "C" int dn_det_loss(const float* cls, const float* labels, const float* loc,
                           const float* targets, const float* mask, long n, int code, float alpha,
                           float gamma, float sigma, float norm, double* losses, float* dcls,
                           float* dloc, void* stream) {
  DN_REQUIRE(cls && labels && loc && targets && mask && losses && dcls && dloc, "det loss: null pointer");
  DN_REQUIRE(n > 0 && code > 0 && norm > 0.f && sigma > 0.f, "det loss: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dn::zero_fill(losses, 2 * sizeof(double), s) != hipSuccess)
    return dn::fail(DN_ERR_LAUNCH, "det loss: memset failed");
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  static const bool legacy = [] { const char* e = getenv("DN_DET_LOSS_LEGACY"); return e && e[0] == '1'; }();
  if (!legacy && n % 2 == 0 && (n * code) % 4 == 0 && al16(cls) && al16(labels) && al16(loc) && al16(targets) && al16(dcls) && al16(dloc))
    hipLaunchKernelGGL(det_loss_v4_kernel, dim3(grid_for(n * code / 4, 4096)), dim3(256), 0, s, cls, labels, loc, targets,
                       mask, n, code, alpha, gamma, sigma, 1.f / norm, losses, dcls, dloc);
  else
    hipLaunchKernelGGL(det_loss_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, s, cls, labels, loc, targets,
                       mask, n, code, alpha, gamma, sigma, 1.f / norm, losses, dcls, dloc);
  return dn::check_launch("det_loss_kernel");
}

extern "C" int dn_adam_step(float* p, const float* g, float* m, float* v, long n, float lr,
                            float beta1, float beta2, float eps, float weight_decay, int step,
                            void* stream) {
  DN_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2));
  return dn::check_launch("adam_kernel");
}

// ---------------------------------------------------------------------------------
// knowledge distillation: KLDivLoss(log_softmax(student), softmax(teacher)) over channels
// ---------------------------------------------------------------------------------
namespace {

// one wavefront per row (= pixel of an NHWC map); mean over ALL elements (the reference's
// nn.KLDivLoss(size_average=True, reduce=True)), times `scale` = kd_weight / (rows * c)
__global__ void __launch_bounds__(256)
kd_kl_kernel(const float* __restrict__ student, const float* __restrict__ teacher, long rows, int c,
             float scale, double* __restrict__ loss, float* __restrict__ dstudent) {
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  for (long row = blockIdx.x * 4L + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4L) {
    const float* s = student + row * c;
    const float* t = teacher + row * c;
    float ms = -INFINITY, mt = -INFINITY;
    for (int k = lane; k < c; k += 64) {
      ms = fmaxf(ms, s[k]);
      mt = fmaxf(mt, t[k]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ms = fmaxf(ms, __shfl_xor(ms, o, 64));
      mt = fmaxf(mt, __shfl_xor(mt, o, 64));
    }
    float es = 0.f, et = 0.f;
    for (int k = lane; k < c; k += 64) {
      es += expf(s[k] - ms);
      et += expf(t[k] - mt);
    }
    es = wave_sum(es);
    et = wave_sum(et);
    const float lse_s = ms + logf(es), lse_t = mt + logf(et);
    float part = 0.f;
    for (int k = lane; k < c; k += 64) {
      const float ls = s[k] - lse_s;              // log_softmax(student)
      const float lt = t[k] - lse_t;              // log softmax(teacher)
      const float pt = expf(lt);
      if (pt > 0.f) part += pt * (lt - ls);
      dstudent[row * c + k] = (expf(ls) - pt) * scale;
    }
    acc += (double)wave_sum(part);
  }
  __shared__ double red[4];
  if (lane == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomic_add_f64(loss, (red[0] + red[1] + red[2] + red[3]) * (double)scale);
}

}  // namespace

extern "C" int dn_kd_kl_loss(const float* student, const float* teacher, long rows, int c, float scale,
                             double* loss, float* dstudent, int zero_loss, void* stream) {
  DN_REQUIRE(student && teacher && loss && dstudent, "kd loss: null pointer");
  DN_REQUIRE(rows > 0 && c > 0, "kd loss: bad shape");
  hipStream_t s = (hipStream_t)stream;
  if (zero_loss && dn::zero_fill(loss, sizeof(double), s) != hipSuccess)
    return dn::fail(DN_ERR_LAUNCH, "kd loss: memset failed");
  const long blocks = (rows + 3) / 4;
  hipLaunchKernelGGL(kd_kl_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s,
                     student, teacher, rows, c, scale, loss, dstudent);
  return dn::check_launch("kd_kl_kernel");
}
