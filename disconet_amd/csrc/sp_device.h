// Device helpers shared by the split-planar conv kernels (conv_sp.hip, conv_spq.hip): vector types, counted
// vmcnt waits, the LDS-DMA instruction, the f16 hi/lo split and the permlane gather that turns MFMA
// accumulator quads into 16-byte SP pieces.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef DN_F32X4_DEFINED
#define DN_F32X4_DEFINED
typedef float f32x4 __attribute__((ext_vector_type(4)));
#endif
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ inline void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N>
__device__ inline void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// x / d and x % d for a work-item index (0 <= x < 2^22, d >= 1) through the float reciprocal the host put into the
// kernel arguments: the work-item decode runs once per tile and runtime integer division costs ~35 instructions
// apiece.  q = trunc(x * rcp) is off by at most one for quotients below 2^20; corrected either way.
__device__ inline int fdivmod(int x, int d, float rcp, int& rem) {
  int q = (int)((float)x * rcp);
  int r = x - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) { ++q; r -= d; }
  rem = r;
  return q;
}

// One LDS-DMA instruction: 64 lanes x 16 bytes, global (per-lane voff + scalar soff) -> LDS at
// lds + 16 * lane (wave-uniform base through M0).  An out-of-range voff writes zeros.  A plain
// __device__ function, not a lambda: the builtin inside a lambda makes hipcc's HOST pass drop the
// kernel's launch stub without a diagnostic.
__device__ inline void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff,
                                           soff, 0, 0);
}

// Sticky range flags of the split-f16 engines (include/disconet_hip.h :: dn_sp_range_flags): bit 1 = a value
// with |x| > 2^14 was split (within two binades of the f16 limit), bit 0 = a value was clamped to +-65504 (the
// result no longer follows the fp32 reference), bit 2 = a NaN reached an epilogue (ReLU / the clamp turn it into a
// finite number: without the flag it would vanish).  One word per translation unit (no relocatable device code in this
// build); dn_sp_range_flags() ORs them.  Written only by lanes that saw such a value: free in the normal case.
__device__ unsigned g_sp_range_flags = 0;

__device__ inline void note_range(float amax, bool nan_seen = false) {
  if (amax > 16384.f) atomicOr(&g_sp_range_flags, amax >= 65504.f ? 3u : 2u);
  if (nan_seen) atomicOr(&g_sp_range_flags, 4u);
}
// NaN test of four values BEFORE a max / clamp can hide them: two unordered compares (true when either operand is a
// NaN), the lane masks OR-ed on the scalar unit -- half a VALU instruction per value.
#ifndef DN_NANCHECK
#define DN_NANCHECK 0      // conv epilogues: 0 = no test (shipped), 1 = every register quad, 2 = quad 0 of every accumulator tile
#endif
__device__ inline void note_nan4(bool& seen, const f32x4 t) {
  seen |= __builtin_isunordered(t[0], t[1]) | __builtin_isunordered(t[2], t[3]);
}
// NaN test of a conv epilogue's pre-activation values.  Measured in one lease (profiles/r04_nancheck_ab.txt): every quad
// costs 1.3 % of the step (3-8 us on each short-K layer), quad 0 of every tile 0.5-1 % -- so the shipped conv epilogues
// carry NONE.  A NaN cannot appear inside the conv stack out of finite operands (overflow is clamped and flagged):
// its sources are the parameters (refused when the plan is packed, model.py :: _check_finite_parameters), the inputs
// (dn_sp_from_nhwc flags them; an occupancy grid cannot hold one) and inf / inf in the attention softmax, which
// dn_disco_fuse_mlp's output test reports (bit 2).
__device__ inline void note_nan4_tile(bool& seen, const f32x4 t, int g) {
  if (DN_NANCHECK == 1 || (DN_NANCHECK == 2 && g == 0)) note_nan4(seen, t);
}

// Stream-ordered readers (no null-stream copy: a hipMemcpyFromSymbol would neither wait for kernels on non-blocking
// streams nor be legal inside a capture).  One thread ORs this translation unit's word into *dst.
__global__ void sp_range_collect_kernel(unsigned* dst, int reset) {
  const unsigned v = g_sp_range_flags;
  if (v) {
    atomicOr(dst, v);
    if (reset) g_sp_range_flags = 0;
  }
}
inline void sp_range_collect_here(unsigned* dst, bool reset, hipStream_t stream) {
  hipLaunchKernelGGL(sp_range_collect_kernel, dim3(1), dim3(1), 0, stream, dst, reset ? 1 : 0);
}

// An LDS read the compiler's wait-count pass does not put behind the LDS-DMA loads in flight.  After a `buffer_load ... lds`
// the pass makes every LDS access it cannot tell apart from the DMA's destination wait for vmcnt(0) -- in a persistent
// kernel that is the NEXT tile's patch, i.e. the epilogue would start only once the prefetch it should hide has landed.
// A load through a __restrict__ parameter carries alias-scope metadata after inlining, and the pass skips scoped loads when
// no DMA store carries a scope (ours do not).  The tables read this way are written once, before the first DMA, behind a
// barrier.  tools: count `s_waitcnt vmcnt(0)` outside ASMSTART blocks in the kernel's ISA.
__device__ __attribute__((always_inline)) inline f32x4 lds_table4(const f32x4* __restrict__ p, const unsigned char* __restrict__ not_p) {
  (void)not_p;
  return *p;
}
#ifndef DN_EPI_NO_DMA_WAIT
#define DN_EPI_NO_DMA_WAIT 1   // tools/ab: 0 = round 3's epilogues (their first affine use waits for the next tile's patch DMA)
#endif
#if !DN_EPI_NO_DMA_WAIT
#define lds_table4(p, q) (*(p))
#endif

// x -> (hi, lo) halves, 4 values -> two dword pairs.  amax: running max |x| of what this lane has split
// (note_range() reports it once per epilogue).
// Vector form on purpose: gfx950 has v_cvt_pk_f16_f32 (two fp32 -> packed halves, round to nearest even, the scalar
// conversion's result) and v_med3_f32 / v_max3_f32 -- 18 VALU instructions per 4 values against ~30 for the
// element-wise form; every epilogue of the short-K layers runs this for each of its outputs.
#ifndef DN_RANGECHECK
#define DN_RANGECHECK 1     // tools/ab: 0 = the splits track no magnitudes (what the range guard costs)
#endif
// lo_clamp: the lower clamp bound -- -65504 (a plain split) or 0 (the ReLU of an epilogue rides in the clamp: max(v, 0) then
// clamp to +-65504 is med3(v, 0, 65504)).  amax is taken from the CLAMPED values: a clamped value is exactly +-65504, which is
// what note_range() reports as "clamped", and anything a ReLU zeroes must not count.
// The lo half: v - float(hi) is exact in fp32; v_fma_mix_f32 reads the f16 half of the packed hi pair directly
// (fma(float(hi), -1, v): the same exact difference) -- one instruction instead of v_cvt_f32_f16 + v_sub_f32.
// DN_SPLIT_V2 = 0 builds round 3's instruction sequence (tools/ab); the values are identical.
#ifndef DN_SPLIT_V2
#define DN_SPLIT_V2 1
#endif
template <int HIGH>
__device__ inline float lo_of_pair(unsigned hpair, float x) {
  float r;
  if constexpr (HIGH) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(x));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(x));
  return r;
}
__device__ inline void split4(const f32x4 v, u32x2& hi, u32x2& lo, float& amax, float lo_clamp = -65504.f) {
  f32x4 x;
#pragma unroll
  for (int e = 0; e < 4; ++e) x[e] = __builtin_amdgcn_fmed3f(v[e], lo_clamp, 65504.f);
#if DN_RANGECHECK
  amax = fmaxf(fmaxf(amax, fabsf(x[0])), fabsf(x[1]));
  amax = fmaxf(fmaxf(amax, fabsf(x[2])), fabsf(x[3]));
#endif
  const half4 h = __builtin_convertvector(x, half4);
  hi = __builtin_bit_cast(u32x2, h);
#if DN_SPLIT_V2
  const f32x4 d = {lo_of_pair<0>(hi[0], x[0]), lo_of_pair<1>(hi[0], x[1]), lo_of_pair<0>(hi[1], x[2]), lo_of_pair<1>(hi[1], x[3])};
  const half4 l = __builtin_convertvector(d, half4);
#else
  const half4 l = __builtin_convertvector(x - __builtin_convertvector(h, f32x4), half4);
#endif
  lo = __builtin_bit_cast(u32x2, l);
}
__device__ inline void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
  float unused = 0.f;
  split4(v, hi, lo, unused);
}
// c * scale + shift of four values as two packed fp32 FMAs (v_pk_fma_f32: the IEEE fma of v_fma_f32, two per instruction)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ inline f32x4 affine4(const f32x4 c, const f32x4 sc, const f32x4 sh) {
#if DN_SPLIT_V2
  const f32x2 a = __builtin_elementwise_fma(f32x2{c[0], c[1]}, f32x2{sc[0], sc[1]}, f32x2{sh[0], sh[1]});
  const f32x2 b = __builtin_elementwise_fma(f32x2{c[2], c[3]}, f32x2{sc[2], sc[3]}, f32x2{sh[2], sh[3]});
  return f32x4{a[0], a[1], b[0], b[1]};
#else
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = c[e] * sc[e] + sh[e];
  return v;
#endif
}
__device__ inline f32x4 quad_of(const f32x16& c, int g) { return f32x4{c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]}; }

// K slices of a conv launch (conv_sp.hip / conv_spq.hip, `KSL` kernels; include/disconet_hip.h :: dn_spconv2d_ks).
// A layer with `count` > 1 canonical K slices defines every output as the fp32 sum, in slice order and starting from
// zero, of the slices' accumulation chains (slice s = K groups [bound(s), bound(s + 1)), each chain from a zeroed
// accumulator).  HOW the slices are distributed does not change a bit of the result: a workgroup that owns a whole
// tile folds them in registers; a tile that is split hands its slices to `count` work items whose raw accumulators
// go to `partial` and a fix-up pass of the same kernel (`fixup`) adds them in the same order.  The launcher splits the
// tiles of the last, under-filled round of a launch (and every tile of a launch smaller than the chip).
struct KSlices {
  float* partial;      // [n_split][count][waves][acc tiles][4 quads][64 lanes] x f32x4
  int count, log2;     // 1, 2 or 4 slices
  int b1, b2, b3;      // group index where slice 1, 2, 3 begins (slice 0 begins at 0, the last ends at ngroups)
  int ngroups;
  int n_whole;         // work items [0, n_whole) are whole tiles
  int n_split;         // tiles behind them, each split into `count` work items
  int fixup;           // this launch only adds the partials of the split tiles and runs their epilogue
  __host__ __device__ int bound(int s) const { return s <= 0 ? 0 : s == 1 ? b1 : s == 2 ? b2 : s == 3 ? b3 : ngroups; }
};

// Lanes (j, 0) and (j, 1) hold channels 4h..4h+3 of octet X (x) and of octet Y (y).  After the
// swaps lane (j, 0) holds octet X complete and lane (j, 1) octet Y complete, as 16 bytes.
__device__ inline u32x4 gather_octet(u32x2 x, u32x2 y) {
  // v_permlane32_swap(a, b): lanes 32-63 of a <-> lanes 0-31 of b
  const auto s0 = __builtin_amdgcn_permlane32_swap(x[0], y[0], false, false);
  const auto s1 = __builtin_amdgcn_permlane32_swap(x[1], y[1], false, false);
  return u32x4{s0[0], s1[0], s0[1], s1[1]};
}

}  // namespace
