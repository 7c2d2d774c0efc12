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

// quad-merged up-conv (conv_spq.hip), reached through dn_spconv2d / dn_spconv_pack_weights (conv_sp.hip)
size_t spq_packed_blocks(int c0g, int nchunks);
int spq_pack_weights(const float* weight_oihw, void* packed, int c_out, int c_in, int c0g, int cout_pad, int nchunks,
                     float wmul, hipStream_t stream);
int spq_conv(const dn_conv_desc* d, const void* src0, const void* src1, const void* packed, size_t packed_bytes,
             const float* scale, const float* shift, void* out, int cout_pad, int bn, hipStream_t stream,
             int kslices = 1, float* workspace = nullptr, size_t workspace_bytes = 0, float* out_nhwc = nullptr,
             int ld_nhwc = 0);

// per-translation-unit words of the split-f16 range flags (sp_device.h); dn_sp_range_flags() ORs them
void range_collect_conv_sp(unsigned* dst, bool reset, hipStream_t stream);
void range_collect_conv_spq(unsigned* dst, bool reset, hipStream_t stream);
void range_collect_fuse_mlp(unsigned* dst, bool reset, hipStream_t stream);
void range_collect_conv_wgrad(unsigned* dst, bool reset, hipStream_t stream);
unsigned* sp_range_word();   // conv_sp.hip's word on the current device (nullptr on error)

}  // namespace dn

#define DN_REQUIRE(cond, ...) \
  do { if (!(cond)) return dn::fail(DN_ERR_ARG, __VA_ARGS__); } while (0)
