// Probe: does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs (needed by the
// split-f16 conv path), and does the x8 form exist on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(float aval, float bval, float* out) {
  half8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.f; b[j] = (_Float16)0.f; }
  // one non-zero slot (lane-half 0, j = 0) on every lane: D[i][j] = a*b for all i, j
  if ((threadIdx.x >> 5) == 0) { a[0] = (_Float16)aval; b[0] = (_Float16)bval; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  half4 a4, b4;
  for (int j = 0; j < 4; ++j) { a4[j] = (_Float16)0.f; b4[j] = (_Float16)0.f; }
  if ((threadIdx.x >> 5) == 0) { a4[0] = (_Float16)aval; b4[0] = (_Float16)bval; }
  f32x16 c8 = {0};
  c8 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c8, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = c8[0]; out[2] = (float)(_Float16)aval; }
}

int main() {
  float* d; hipMalloc(&d, 16);
  const float tests[][2] = {{1e-6f, 1.0f}, {1.0f, 1e-6f}, {3e-5f, 3e-5f}, {0.5f, 0.25f}, {6e-8f, 1024.f}};
  for (auto& t : tests) {
    probe<<<1, 64>>>(t[0], t[1], d);
    float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("a=%g b=%g  x16: %.9g  x8: %.9g  (half(a)=%.9g, exact a*b=%.9g)\n", t[0], t[1], h[0], h[1], h[2], (double)t[0] * t[1]);
  }
  return 0;
}
