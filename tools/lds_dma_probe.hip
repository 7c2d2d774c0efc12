// Probe: buffer_load ... lds (LDS-DMA) with 16-byte lanes on gfx950 -- does an
// out-of-range lane write zeros to LDS, and is the destination lane-linear?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* src, int nbytes, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 2];
  for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) lds[i] = -7.f;   // poison
  __syncthreads();
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
  // lane L fetches source float4 #(63 - L) (reversed), lanes 8..15 out of range
  unsigned voff = (63 - threadIdx.x) * 16;
  if (threadIdx.x >= 8 && threadIdx.x < 16) voff = 0xFFFFFFFFu;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 4; i += 64) out[i] = lds[i];
}

int main() {
  float h[256], *d, *o;
  for (int i = 0; i < 256; ++i) h[i] = (float)i;
  hipMalloc(&d, 1024); hipMalloc(&o, 1024);
  hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d, 1024, o);
  float r[256];
  hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
  printf("hip error: %s\n", hipGetErrorString(hipGetLastError()));
  for (int lane : {0, 1, 7, 8, 15, 16, 63})
    printf("LDS slot %2d: %6.1f %6.1f %6.1f %6.1f   (expect source float4 #%d%s)\n", lane, r[4 * lane],
           r[4 * lane + 1], r[4 * lane + 2], r[4 * lane + 3], 63 - lane,
           (lane >= 8 && lane < 16) ? " -> zeros if OOB writes 0" : "");
  return 0;
}
