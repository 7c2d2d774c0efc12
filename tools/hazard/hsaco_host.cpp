// Loads one assembled code object of the round-1 warp kernel (asm_edit.py) and launches it.
//   hipcc -O2 -shared -fPIC tools/hazard/hsaco_host.cpp -o tools/hazard/hsaco_host.so
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

struct Args {                     // the kernel's argument block (tools/hazard_warp_r1.hip :: warp_neighbors_kernel)
  const float* feat; const float* trans; const int32_t* num_agent;
  int batch, agents, h, w, c, only_v2i, ego_first, ego_count;
  float* warped; unsigned* trace; unsigned* dbg;
};

extern "C" void* hz_load(const char* path, const char* kernel) {
  hipModule_t mod;
  hipFunction_t fn;
  if (hipModuleLoad(&mod, path) != hipSuccess) { fprintf(stderr, "hipModuleLoad(%s) failed\n", path); return nullptr; }
  if (hipModuleGetFunction(&fn, mod, kernel) != hipSuccess) { fprintf(stderr, "no kernel %s\n", kernel); return nullptr; }
  return (void*)fn;
}

extern "C" int hz_launch(void* fn, const float* feat, const float* trans, const int32_t* num_agent, int batch, int agents,
                         int h, int w, int c, float* warped, unsigned* trace, void* stream) {
  Args a{feat, trans, num_agent, batch, agents, h, w, c, 0, 0, agents, warped, trace, nullptr};
  size_t size = sizeof(a);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  return (int)hipModuleLaunchKernel((hipFunction_t)fn, (h * w + 31) / 32, agents - 1, batch * agents, 256, 1, 1, 0,
                                    (hipStream_t)stream, nullptr, extra);
}
