/* A C host of libdisconet_hip.so with no Python and no torch: the drop-in boundary
 * is the C ABI of include/disconet_hip.h.
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/c_abi/check_abi.c \
 *       -L disconet_amd -ldisconet_hip -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,... -o check_abi
 *   ./check_abi          error behaviour only (no GPU needed)
 *   ./check_abi --gpu    + a 3x3 conv (both math modes) and the voxelizer against C loops
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "disconet_hip.h"
#include "disconet_train.h"

#define CHECK(cond, ...) do { if (!(cond)) { printf("FAIL line %d: ", __LINE__); printf(__VA_ARGS__); printf("\n"); return 1; } } while (0)
#define HIP(x) CHECK((x) == hipSuccess, "%s", #x)

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) / 8388608.0f) - 1.0f; }

static int errors_only(void) {
  CHECK(dn_version() >= 100, "version");
  dn_conv_desc d;
  memset(&d, 0, sizeof d);
  d.n_images = 1; d.h_in = 8; d.w_in = 8; d.c0 = 32; d.c_out = 32; d.ksize = 7; d.stride = 1;
  d.ld0 = 32; d.ldo = 32;
  CHECK(dn_conv_packed_weight_floats(&d) == 0, "bad ksize must give 0 floats");
  CHECK(strstr(dn_last_error(), "ksize") != NULL, "message: %s", dn_last_error());
  CHECK(dn_conv2d(&d, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == DN_ERR_ARG, "conv arg error");
  d.ksize = 3; d.math = 5;
  CHECK(dn_conv2d(&d, NULL, NULL, NULL, NULL, NULL, NULL, NULL) == DN_ERR_ARG, "bad math mode");
  CHECK(dn_decode_boxes(NULL, NULL, NULL, 1, 10, NULL, NULL, NULL) == DN_ERR_ARG, "decode null");
  int dims[3] = {256, 256, 13};
  CHECK(dn_voxel_compact_workspace(dims) == ((256 * 256 * 13 + 1023) / 1024) * sizeof(int), "workspace");
  CHECK(dn_post1x1_packed_floats() == 64 * 64, "post1x1 size");
  /* training header: plain C as well, same error convention */
  d.math = 0;
  CHECK(dn_conv_wgrad_workspace(&d) > 0, "wgrad workspace");
  CHECK(dn_conv_wgrad(&d, NULL, NULL, NULL, NULL, NULL, 0, 0, NULL) == DN_ERR_ARG, "wgrad null");
  CHECK(dn_adam_step(NULL, NULL, NULL, NULL, 10, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.f, 1, NULL) == DN_ERR_ARG,
        "adam null");
  CHECK(dn_bn_train_stats(NULL, 1, 10, 600, 600, NULL, NULL, NULL, NULL) == DN_ERR_ARG, "bn null");
  printf("C ABI error behaviour: ok\n");
  return 0;
}

static int gpu_conv(int math) {
  const int n = 2, h = 12, w = 20, cin = 32, cout = 48;
  dn_conv_desc d;
  memset(&d, 0, sizeof d);
  d.n_images = n; d.h_in = h; d.w_in = w; d.c0 = cin; d.c_out = cout; d.ksize = 3; d.stride = 1;
  d.relu = 1; d.ld0 = cin; d.ldo = cout; d.math = math;
  const size_t nx = (size_t)n * h * w * cin, ny = (size_t)n * h * w * cout, nw = (size_t)cout * cin * 9;
  float *x = malloc(nx * 4), *wt = malloc(nw * 4), *bias = malloc(cout * 4), *y = malloc(ny * 4), *ref = malloc(ny * 4);
  unsigned s = 7;
  for (size_t i = 0; i < nx; ++i) x[i] = frand(&s);
  for (size_t i = 0; i < nw; ++i) wt[i] = frand(&s) * 0.08f;
  for (int i = 0; i < cout; ++i) bias[i] = frand(&s) * 0.1f;
  for (int im = 0; im < n; ++im) for (int oy = 0; oy < h; ++oy) for (int ox = 0; ox < w; ++ox) for (int co = 0; co < cout; ++co) {
    double acc = bias[co];
    for (int dy = 0; dy < 3; ++dy) for (int dx = 0; dx < 3; ++dx) {
      const int iy = oy + dy - 1, ix = ox + dx - 1;
      if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
      for (int ci = 0; ci < cin; ++ci)
        acc += (double)x[((size_t)(im * h + iy) * w + ix) * cin + ci] * wt[((size_t)co * cin + ci) * 9 + dy * 3 + dx];
    }
    ref[((size_t)(im * h + oy) * w + ox) * cout + co] = acc > 0 ? (float)acc : 0.f;
  }
  float *dx_, *dw, *db, *dp, *dsc, *dsh, *dy_;
  const size_t np = dn_conv_packed_weight_floats(&d);
  CHECK(np > 0, "packed size: %s", dn_last_error());
  HIP(hipMalloc((void**)&dx_, nx * 4)); HIP(hipMalloc((void**)&dw, nw * 4)); HIP(hipMalloc((void**)&db, cout * 4));
  HIP(hipMalloc((void**)&dp, np * 4)); HIP(hipMalloc((void**)&dsc, cout * 4)); HIP(hipMalloc((void**)&dsh, cout * 4));
  HIP(hipMalloc((void**)&dy_, ny * 4));
  HIP(hipMemcpy(dx_, x, nx * 4, hipMemcpyHostToDevice)); HIP(hipMemcpy(dw, wt, nw * 4, hipMemcpyHostToDevice));
  HIP(hipMemcpy(db, bias, cout * 4, hipMemcpyHostToDevice));
  CHECK(dn_conv_pack_weights(&d, dw, dp, NULL) == DN_OK, "pack: %s", dn_last_error());
  CHECK(dn_fold_bn(db, NULL, NULL, NULL, NULL, 0.f, cout, dsc, dsh, NULL) == DN_OK, "fold: %s", dn_last_error());
  CHECK(dn_conv2d(&d, dx_, NULL, dp, dsc, dsh, dy_, NULL) == DN_OK, "conv: %s", dn_last_error());
  HIP(hipDeviceSynchronize());
  HIP(hipMemcpy(y, dy_, ny * 4, hipMemcpyDeviceToHost));
  double err = 0;
  for (size_t i = 0; i < ny; ++i) { const double e = fabs((double)y[i] - ref[i]); if (e > err) err = e; }
  printf("C ABI conv3x3 math=%d: max abs err %.3e\n", math, err);
  CHECK(err <= 1e-4, "conv error too large");
  return 0;
}

static int gpu_voxel(void) {
  const int npts = 5000;
  float* pts = malloc((size_t)npts * 4 * 4);
  unsigned s = 3;
  for (int i = 0; i < npts; ++i) {
    pts[4 * i] = frand(&s) * 33.f; pts[4 * i + 1] = frand(&s) * 33.f; pts[4 * i + 2] = frand(&s) * 3.f - 0.5f; pts[4 * i + 3] = 0.f;
  }
  const double vs[3] = {0.25, 0.25, 0.4}, ext[6] = {-32, 32, -32, 32, -3, 2};
  const int dims[3] = {256, 256, 13};
  const size_t cells = 256 * 256 * 13;
  float* ref = calloc(cells, 4);
  for (int i = 0; i < npts; ++i) {
    const double x = pts[4 * i], y = pts[4 * i + 1], z = pts[4 * i + 2];
    if (!(ext[0] < x && x < ext[1] && ext[2] < y && y < ext[3] && ext[4] < z && z < ext[5])) continue;
    const int qx = (int)floor(x / vs[0]) + 128, qy = (int)floor(y / vs[1]) + 128, qz = (int)floor(z / vs[2]) + 8;
    ref[((size_t)qx * 256 + qy) * 13 + qz] = 1.f;
  }
  float *dp, *dd, *out = malloc(cells * 4);
  HIP(hipMalloc((void**)&dp, (size_t)npts * 16)); HIP(hipMalloc((void**)&dd, cells * 4));
  HIP(hipMemcpy(dp, pts, (size_t)npts * 16, hipMemcpyHostToDevice));
  CHECK(dn_voxelize_occupy(dp, npts, 4, vs, ext, dims, dd, NULL) == DN_OK, "voxelize: %s", dn_last_error());
  HIP(hipDeviceSynchronize());
  HIP(hipMemcpy(out, dd, cells * 4, hipMemcpyDeviceToHost));
  CHECK(memcmp(out, ref, cells * 4) == 0, "voxel grids differ");
  printf("C ABI voxelizer: bit-exact\n");
  return 0;
}

int main(int argc, char** argv) {
  if (errors_only()) return 1;
  if (argc > 1 && strcmp(argv[1], "--gpu") == 0) {
    if (gpu_conv(0) || gpu_conv(1) || gpu_voxel()) return 1;
  }
  printf("OK\n");
  return 0;
}
