// Shared helpers of libdisconet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "disconet_hip.h"

namespace dn {

char* err_buf();   // thread-local, 512 bytes

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

// Zero `bytes` bytes at `ptr` on `stream` with a kernel launch.  NOT hipMemsetAsync: a memset captured into a
// hipGraph was measured to write a garbage 16-byte pattern from the second replay on (tools/hazard/ptr_audit.py,
// DESIGN.md 3.6) -- a kernel node carries its arguments by value and replays exactly.
hipError_t zero_fill(void* ptr, size_t bytes, hipStream_t stream);

// Launch-time caches (kernel attributes) are per DEVICE: a process that drives several GPUs must set the
// dynamic-LDS attribute on each.  `flags` is a function-local static of the caller.
struct PerDeviceFlag {
  bool done[64] = {};
  bool& here() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return done[dev & 63];
  }
};

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(DN_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return DN_OK;
}

}  // namespace dn

#define DN_REQUIRE(cond, ...) \
  do { if (!(cond)) return dn::fail(DN_ERR_ARG, __VA_ARGS__); } while (0)
