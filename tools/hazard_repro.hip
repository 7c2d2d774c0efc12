// Minimal reproducer for DESIGN.md §3.6: a kernel with a LANE-DEPENDENT loop exit
//     for (int c4 = lane; c4 < c4n; c4 += 64) { ... store row[c4] ... }
// whose waves share SIMDs with the LDS-read + MFMA dense loop of another kernel on a second stream.
// Round 1 saw lanes 48..63 of the warp kernel run one extra iteration (c4 = lane + 64: the NEXT
// pixel's channels 192..255) in that situation.  This tool keeps the two kernels at < 100 lines,
// runs the victim alone (reference) and beside the aggressor, and counts launches whose output
// differs, for the original loop form and for the candidate fixes:
//   variant 0  original lane-dependent exit
//   variant 1  wave-uniform trip count, no per-lane condition for whole rows (the shipped fix)
//   variant 2  original exit + s_nop 15 at the loop top and after the store
//   variant 3  original exit + s_waitcnt vmcnt(0) lgkmcnt(0) at the loop bottom
//   variant 4  original exit, body without the 16 tap loads (stores only)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hazard_repro.hip -o tools/hazard_repro.bin [-save-temps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int PIX_PER_BLOCK = 32, HW = 1024, C = 256;

template <int VARIANT>
__global__ void __launch_bounds__(256) victim(const float* __restrict__ src, float* __restrict__ dst, int c) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c4n = c >> 2;
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src + (size_t)blockIdx.y * HW * c), 0, HW * c * 4, 0x00020000);
  float* img = dst + (size_t)blockIdx.y * HW * c;
  for (int pp = wave; pp < PIX_PER_BLOCK; pp += 4) {
    const int p = blockIdx.x * PIX_PER_BLOCK + pp;
    float* out = img + (size_t)p * c;
    int tap[16];
    float wgt[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { tap[k] = (p * 7 + k * 37 + blockIdx.y * 11) & (HW - 1); wgt[k] = 0.0625f * (1 + (k & 3)); }
    auto body = [&](int c4) {
      f32x4 acc = {(float)p, 0.f, 0.f, 0.f};
      if (VARIANT != 4) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
          acc += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)(tap[k] * c * 4 + 16 * c4), 0, 0)) * wgt[k];
      }
      *reinterpret_cast<f32x4*>(out + 4 * c4) = acc;
    };
    if (VARIANT == 1) {
      const int n_it = (c4n + 63) >> 6;
      const bool full = (c4n & 63) == 0;
      for (int it = 0; it < n_it; ++it)
        if (full || lane + 64 * it < c4n) body(lane + 64 * it);
    } else {
      for (int c4 = lane; c4 < c4n; c4 += 64) {
        if (VARIANT == 2) asm volatile("s_nop 15");
        body(c4);
        if (VARIANT == 2) asm volatile("s_nop 15");
        if (VARIANT == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
    }
  }
}

// the conv engine's inner loop, stripped: ds_read_b128 x4 + v_mfma_f32_32x32x16_f16 x12 per trip
__global__ void __launch_bounds__(256) aggressor(float* sink, int trips) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<float*>(lds)[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x16 acc[4] = {};
  const unsigned char* base = lds + (threadIdx.x & 63) * 16;
  for (int t = 0; t < trips; ++t) {
    const int off = (t & 15) * 4096;
    const half8 a0 = *reinterpret_cast<const half8*>(base + off), a1 = *reinterpret_cast<const half8*>(base + off + 1024);
    const half8 b0 = *reinterpret_cast<const half8*>(base + off + 2048), b1 = *reinterpret_cast<const half8*>(base + off + 3072);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a0, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a1, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a1, acc[3], 0, 0, 0);
    }
  }
  if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.f) sink[threadIdx.x] = 1.f;
}

template <int VARIANT>
void run(const char* name, const float* src, float* out, float* ref, int pairs, float* sink, hipStream_t s1, hipStream_t s2) {
  const size_t n = (size_t)pairs * HW * C;
  std::vector<float> hr(n), ho(n);
  hipMemsetAsync(ref, 0, n * 4, s1);
  victim<VARIANT><<<dim3(HW / PIX_PER_BLOCK, pairs), 256, 0, s1>>>(src, ref, C);
  hipStreamSynchronize(s1);
  hipMemcpy(hr.data(), ref, n * 4, hipMemcpyDeviceToHost);
  int bad_alone = 0, bad_beside = 0;
  long first = -1;
  for (int mode = 0; mode < 2; ++mode)
    for (int trial = 0; trial < 16; ++trial) {
      hipMemsetAsync(out, 0, n * 4, s1);
      hipStreamSynchronize(s1);
      if (mode) {
        aggressor<<<1024, 256, 65536, s2>>>(sink, 4000);
        aggressor<<<1024, 256, 65536, s2>>>(sink, 4000);
      }
      victim<VARIANT><<<dim3(HW / PIX_PER_BLOCK, pairs), 256, 0, s1>>>(src, out, C);
      hipDeviceSynchronize();
      hipMemcpy(ho.data(), out, n * 4, hipMemcpyDeviceToHost);
      if (memcmp(ho.data(), hr.data(), n * 4)) {
        (mode ? bad_beside : bad_alone)++;
        if (first < 0)
          for (size_t i = 0; i < n; ++i)
            if (memcmp(&ho[i], &hr[i], 4)) { first = (long)i; break; }
      }
    }
  printf("variant %d %-58s alone: %2d of 16 differ   beside the MFMA loop: %2d of 16 differ", VARIANT, name, bad_alone, bad_beside);
  if (first >= 0) printf("   first: image %ld pixel %ld channel %ld", first / ((long)HW * C), (first / C) % HW, first % C);
  printf("\n");
}

int main() {
  const int pairs = 80;
  const size_t n = (size_t)pairs * HW * C;
  float *src, *out, *ref, *sink;
  hipMalloc(&src, n * 4); hipMalloc(&out, n * 4); hipMalloc(&ref, n * 4); hipMalloc(&sink, 4096);
  std::vector<float> h(n);
  unsigned r = 1;
  for (auto& x : h) { r = r * 1664525u + 1013904223u; x = (r >> 8) / 8388608.0f - 1.0f; }
  hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipStream_t s1, s2;
  hipStreamCreate(&s1); hipStreamCreate(&s2);
  run<0>("original: for (c4 = lane; c4 < c4n; c4 += 64)", src, out, ref, pairs, sink, s1, s2);
  run<1>("wave-uniform trip count (shipped fix)", src, out, ref, pairs, sink, s1, s2);
  run<2>("original + s_nop 15 around the body", src, out, ref, pairs, sink, s1, s2);
  run<3>("original + s_waitcnt vmcnt(0) lgkmcnt(0) at the bottom", src, out, ref, pairs, sink, s1, s2);
  run<4>("original, stores only (no tap loads)", src, out, ref, pairs, sink, s1, s2);
  printf("hip error state: %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
