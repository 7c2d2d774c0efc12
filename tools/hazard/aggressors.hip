// Single-ingredient "other stream" kernels for tools/hazard/corun_asm.py: each loops over ONE instruction kind that
// the split-f16 conv engines issue and the fp32 engine / the library GEMMs do not, so that the kernel whose neighbour
// (the round-1 warp kernel, e0.hsaco) goes wrong names the ingredient.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/hazard/aggressors.hip -o tools/hazard/aggressors.so
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void __launch_bounds__(256) aggressor(float* sink, int trips) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];     // 64 KiB: two workgroups per CU, like the conv
  float a = threadIdx.x * 0.37f + 1.f, b = a * 1.7f, c = 0.f, d = 0.f;
  unsigned u = threadIdx.x, v = u * 3u;
  f32x16 acc = {};
  half8 h = {1, 2, 3, 4, 5, 6, 7, 8};
  if (KIND == 5) { reinterpret_cast<float*>(lds)[threadIdx.x] = a; __syncthreads(); }
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      if (KIND == 0) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u) : "v"(a), "v"(b));
      if (KIND == 1) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(c) : "v"(u));
      if (KIND == 2) asm volatile("v_cvt_f16_f32_e32 %0, %1\n\tv_cvt_f32_f16_e32 %2, %0" : "=&v"(u), "+v"(a), "=v"(c));
      if (KIND == 3) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(v));
      if (KIND == 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, acc, 0, 0, 0);
      if (KIND == 5) { asm volatile("ds_read_b128 %0, %1" : "=v"(h) : "v"((threadIdx.x & 63) * 16) : "memory"); }
      if (KIND == 6) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*reinterpret_cast<double*>(&c)) : "v"(*reinterpret_cast<double*>(&a)), "v"(*reinterpret_cast<double*>(&a)));
      if (KIND == 7) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(u) : "v"(v), "v"(0xffffu));
    }
    if (KIND == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (a + b + c + d + acc[0] + (float)h[0] + (float)u + (float)v == 12345.678f) sink[threadIdx.x] = 1.f;
}

static float* g_sink = nullptr;
template <int KIND> static int run(int trips, hipStream_t s) {
  if (!g_sink) hipMalloc(&g_sink, 4096);
  hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(aggressor<KIND>, dim3(1024), dim3(256), 65536, s, g_sink, trips);
  return (int)hipGetLastError();
}
extern "C" int hz_aggressor(int kind, int trips, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (kind) {
    case 0: return run<0>(trips, s); case 1: return run<1>(trips, s); case 2: return run<2>(trips, s);
    case 3: return run<3>(trips, s); case 4: return run<4>(trips, s); case 5: return run<5>(trips, s);
    case 6: return run<6>(trips, s); case 7: return run<7>(trips, s);
  }
  return -1;
}
