// Ablation harness for conv_mfma_kernel (not part of the product library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I disconet_amd/csrc \
//         tools/conv_ablate.hip disconet_amd/csrc/common.hip -o /tmp/conv_ablate
// Times one layer shape with the normal kernel (ABL 0) and the ablation
// variants (see the ABL comment in conv_mfma.hip) to show where the gap to the
// fp32-MFMA roof goes.
#include <cstdio>
#include <vector>
#include "../disconet_amd/csrc/conv_mfma.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KS, int S, int TH, int TW, int BN, int KC, int WM, int WN, int WTM, int WTN, int ABL>
float time_variant(ConvArgs a, const dn_conv_desc& d, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch<KS, S, TH, TW, BN, KC, WM, WN, WTM, WTN, ABL>(a, d, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch<KS, S, TH, TW, BN, KC, WM, WN, WTM, WTN, ABL>(a, d, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

template <int KS, int S, int TH, int TW, int BN, int KC, int WM, int WN, int WTM, int WTN>
int run_shape(const char* name, int n, int h, int w, int c0, int c1, int up0, int cout) {
  dn_conv_desc d = {n, h, w, c0, c1, up0, cout, KS, S, 1, c0, c1, cout};
  const int ho = out_dim(h, KS, S), wo = out_dim(w, KS, S);
  const size_t n0 = (size_t)n * (up0 ? (h / 2) * (w / 2) : h * w) * c0, n1 = (size_t)n * h * w * (c1 ? c1 : 1);
  const size_t no = (size_t)n * ho * wo * cout, nw = dn_conv_packed_weight_floats(&d);
  float *s0, *s1, *wp, *sc, *sh, *out;
  CK(hipMalloc(&s0, n0 * 4)); CK(hipMalloc(&s1, n1 * 4)); CK(hipMalloc(&wp, nw * 4));
  CK(hipMalloc(&sc, cout * 4)); CK(hipMalloc(&sh, cout * 4)); CK(hipMalloc(&out, no * 4));
  std::vector<float> hbuf(std::max(std::max(n0, n1), nw));
  unsigned r = 12345;
  auto fill = [&](float* dptr, size_t cnt, float scale) {
    for (size_t i = 0; i < cnt; ++i) { r = r * 1664525u + 1013904223u; hbuf[i] = ((r >> 8) / 8388608.0f - 1.0f) * scale; }
    return hipMemcpy(dptr, hbuf.data(), cnt * 4, hipMemcpyHostToDevice);
  };
  CK(fill(s0, n0, 1.f)); CK(fill(s1, n1, 1.f)); CK(fill(wp, nw, 0.05f)); CK(fill(sc, cout, 1.f)); CK(fill(sh, cout, 0.1f));
  ConvArgs a;
  if (fill_args(&d, s0, c1 ? s1 : nullptr, wp, sc, sh, out, a) != DN_OK) { printf("%s\n", dn_last_error()); return 1; }
  const double flop = 2.0 * n * ho * wo * cout * (c0 + c1) * KS * KS;
  const int iters = 20;
  float t[8];
  g_persist = 0;
  const float t_np = time_variant<KS, S, TH, TW, BN, KC, WM, WN, WTM, WTN, 0>(a, d, iters);
  g_persist = 2;   // persistent workgroups for every variant below
  t[0] = time_variant<KS, S, TH, TW, BN, KC, WM, WN, WTM, WTN, 0>(a, d, iters);
  t[1] = time_variant<KS, S, TH, TW, BN, KC, WM, WN, WTM, WTN, 1>(a, d, iters);
  t[2] = 0;
  t[3] = time_variant<KS, S, TH, TW, BN, KC, WM, WN, WTM, WTN, 3>(a, d, iters);
  t[4] = 0;
  t[5] = time_variant<KS, S, TH, TW, BN, KC, WM, WN, WTM, WTN, 5>(a, d, iters);
  t[6] = 0; t[7] = 0;
  const char* lab[8] = {"normal", "no-stream", "no-stream+no-store", "no-store", "no-lds-reads", "pure-mfma",
                        "global-loads-only", "lds-store+barrier-only"};
  printf("%s  tile %dx%d  (%.2f GFLOP)\n", name, TH * TW, BN, flop / 1e9);
  printf("   ---- one workgroup per item    %8.1f us  %7.1f TFLOP/s\n", t_np * 1e3, flop / (t_np * 1e-3) / 1e12);
  for (int i = 0; i < 8; ++i) if (t[i] > 0) printf("   ABL%d %-20s %8.1f us  %7.1f TFLOP/s\n", i, lab[i], t[i] * 1e3, flop / (t[i] * 1e-3) / 1e12);
  hipFree(s0); hipFree(s1); hipFree(wp); hipFree(sc); hipFree(sh); hipFree(out);
  return 0;
}

int main() {
  //               KS S TH TW  BN  KC WM WN WTM WTN
  if (run_shape<3, 1, 8, 32, 32, 16, 4, 1, 2, 1>("conv8_2  20x256x256  32->32 ", 20, 256, 256, 32, 0, 0, 32)) return 1;
  if (run_shape<3, 1, 8, 32, 64, 16, 4, 1, 2, 2>("conv7_2  20x128x128  64->64 ", 20, 128, 128, 64, 0, 0, 64)) return 1;
  if (run_shape<3, 1, 8, 16, 64, 16, 2, 2, 2, 1>("conv6_2  20x64x64  128->128 ", 20, 64, 64, 128, 0, 0, 128)) return 1;
  if (run_shape<3, 1, 8, 8, 64, 16, 2, 2, 1, 1>("conv5_2  20x32x32  256->256 ", 20, 32, 32, 256, 0, 0, 256)) return 1;
  if (run_shape<3, 1, 8, 8, 64, 16, 2, 2, 1, 1>("conv5_1  20x32x32  768->256 up+cat", 20, 32, 32, 512, 256, 1, 256)) return 1;
  if (run_shape<3, 1, 8, 16, 128, 8, 2, 2, 2, 2>("conv5_1  same, 128x128 tile", 20, 32, 32, 512, 256, 1, 256)) return 1;
  return 0;
}
