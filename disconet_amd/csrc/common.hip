#include "dn_internal.h"

namespace dn {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
namespace {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// head/tail bytes one at a time, the 16-byte aligned middle as 128-bit stores (grid-stride)
__global__ void __launch_bounds__(256) zero_fill_kernel(unsigned char* p, size_t head, size_t vecs, size_t tail) {
  const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  if (tid < head) p[tid] = 0;
  u32x4* v = reinterpret_cast<u32x4*>(p + head);
  for (size_t i = tid; i < vecs; i += stride) v[i] = u32x4{0u, 0u, 0u, 0u};
  if (tid < tail) p[head + 16 * vecs + tid] = 0;
}
}  // namespace

hipError_t zero_fill(void* ptr, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return hipSuccess;
  unsigned char* p = static_cast<unsigned char*>(ptr);
  size_t head = (16 - (reinterpret_cast<size_t>(p) & 15)) & 15;
  if (head > bytes) head = bytes;
  const size_t vecs = (bytes - head) / 16, tail = bytes - head - 16 * vecs;
  const size_t want = (vecs + 255) / 256;
  const int blocks = (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
  hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks), dim3(256), 0, stream, p, head, vecs, tail);
  return hipGetLastError();
}
}  // namespace dn

extern "C" int dn_version(void) { return 134; }  // 0.1.3: round-5 kernels (profiles carry this number; 131: dn_conv_wgrad_sp; 132: round 6 -- Gray MFMA order, fp32 rows from the tap-merged kernel, fused bias gradients; 133: dn_spconv_pack_weights_multi, lane-parallel warp gathers; 134: dn_conv_wgrad_sp_z, dz = NULL in the BatchNorm backward)

// The hash of every source / header / flag this library was built from (csrc/build.py :: tree_hash), behind a marker
// that build.py also finds in the file without loading it.  _lib.load() refuses a library whose id is not the tree's.
#ifndef DN_BUILD_ID
#error "build through disconet_amd/csrc/build.py (it passes -DDN_BUILD_ID)"
#endif
static const char dn_build_id_marker[] = "dn-build-id:" DN_BUILD_ID;
extern "C" const char* dn_build_id(void) { return dn_build_id_marker + 12; }

// Stream-ordered form: every split-f16 translation unit ORs its sticky word into *dst_device (which the caller
// zeroed) behind whatever the stream holds; kernel launches only, so it is legal inside a capture.
extern "C" int dn_sp_range_flags_async(unsigned* dst_device, int reset, void* stream) {
  DN_REQUIRE(dst_device != nullptr, "sp_range_flags_async: null destination");
  hipStream_t s = (hipStream_t)stream;
  // bit 1 of `reset`: zero *dst first, with a kernel -- a captured step cannot rely on a memset node
  if ((reset & 2) && dn::zero_fill(dst_device, sizeof(unsigned), s) != hipSuccess)
    return dn::fail(DN_ERR_LAUNCH, "sp_range_flags_async: zero fill of the destination failed");
  const bool clear = (reset & 1) != 0;
  dn::range_collect_conv_sp(dst_device, clear, s);
  dn::range_collect_conv_spq(dst_device, clear, s);
  dn::range_collect_fuse_mlp(dst_device, clear, s);
  dn::range_collect_conv_wgrad(dst_device, clear, s);
  return dn::check_launch("sp_range_collect_kernel");
}

// Blocking form: waits for the device (every stream, blocking or not), then reads through the same collectors.
extern "C" unsigned dn_sp_range_flags(int reset) {
  unsigned* d = nullptr;
  unsigned v = 0;
  if (hipDeviceSynchronize() != hipSuccess || hipMalloc((void**)&d, sizeof v) != hipSuccess) return 0x80000000u;
  bool ok = hipMemcpy(d, &v, sizeof v, hipMemcpyHostToDevice) == hipSuccess &&
            dn_sp_range_flags_async(d, reset ? 1 : 0, nullptr) == DN_OK &&
            hipMemcpy(&v, d, sizeof v, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d);
  return ok ? v : 0x80000000u;
}

extern "C" const char* dn_last_error(void) { return dn::err_buf(); }
