// Tile sweep for conv_mfma_kernel (not part of the product library):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I disconet_amd/csrc \
//         tools/conv_tile_sweep.hip disconet_amd/csrc/common.hip -o tools/conv_tile_sweep.bin
// Times every 3x3 stride-1 tile of the menu on the layer shapes of the BASELINE
// workload, in both math modes, to ground select_cfg()'s preference order.
#include <cstdio>
#include <vector>
#include "../disconet_amd/csrc/conv_mfma.hip"

struct Bufs { float *s0, *s1, *wp, *sc, *sh, *out; };

template <int KS, int S, int TH, int TW, int BN, int KC, int WM, int WN, int WTM, int WTN, int MATH, int ABL = 0>
float time_tile(ConvArgs a, const dn_conv_desc& d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch<KS, S, TH, TW, BN, KC, WM, WN, WTM, WTN, ABL, MATH>(a, d, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int iters = 10;
  for (int i = 0; i < iters; ++i) launch<KS, S, TH, TW, BN, KC, WM, WN, WTM, WTN, ABL, MATH>(a, d, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters * 1e3f;
}

template <int MATH>
void sweep(const char* name, int n, int h, int w, int c0, int c1, int up0, int cout, int stride, Bufs b) {
  dn_conv_desc d = {n, h, w, c0, c1, up0, cout, 3, stride, 1, c0, c1, cout, MATH};
  ConvArgs a;
  a.src0 = b.s0; a.src1 = c1 ? b.s1 : nullptr; a.wpk = b.wp; a.scale = b.sc; a.shift = b.sh; a.out = b.out;
  a.n_images = n; a.h_in = h; a.w_in = w; a.h_out = out_dim(h, 3, stride); a.w_out = out_dim(w, 3, stride);
  a.c0 = c0; a.c1 = c1; a.up0 = up0; a.c_out = cout; a.relu = 1; a.ld0 = c0; a.ld1 = c1; a.ldo = cout;
  a.cout_pad = cout_pad_of(d); a.vec0 = 1; a.vec1 = c1 ? 1 : 0; a.vec_out = 1;
  a.wpk_bytes = (int)(dn_conv_packed_weight_floats(&d) * 4);
  const double gf = 2.0 * n * a.h_out * a.w_out * cout * (c0 + c1) * 9 / 1e9;
  printf("%-34s math %d  %6.2f GF :", name, MATH, gf);
  if (stride == 1) {
    float t;
    t = time_tile<3, 1, 8, 32, 32, 16, 4, 1, 2, 1, MATH>(a, d);  printf("  256x32 %6.1f", t);
    t = time_tile<3, 1, 8, 32, 64, 16, 4, 1, 2, 2, MATH>(a, d);  printf("  256x64 %6.1f", t);
    t = time_tile<3, 1, 8, 16, 128, 8, 2, 2, 2, 2, MATH>(a, d);  printf("  128x128 %6.1f", t);
    t = time_tile<3, 1, 8, 16, 64, 16, 2, 2, 2, 1, MATH>(a, d);  printf("  128x64 %6.1f", t);
    t = time_tile<3, 1, 8, 8, 64, 16, 2, 2, 1, 1, MATH>(a, d);   printf("  64x64 %6.1f", t);
    if (MATH == 1) {
      t = time_tile<3, 1, 8, 32, 32, 16, 4, 1, 2, 1, 1, 8>(a, d);  printf("  | no-split-VALU: 256x32 %6.1f", t);
      t = time_tile<3, 1, 8, 32, 64, 16, 4, 1, 2, 2, 1, 8>(a, d);  printf("  256x64 %6.1f", t);
      t = time_tile<3, 1, 8, 32, 32, 16, 4, 1, 2, 1, 1, 1>(a, d);  printf("  | no-stream: 256x32 %6.1f", t);
      t = time_tile<3, 1, 8, 32, 32, 16, 4, 1, 2, 1, 1, 3>(a, d);  printf("  | no-store: 256x32 %6.1f", t);
      t = time_tile<3, 1, 8, 32, 32, 16, 4, 1, 2, 1, 1, 9>(a, d);  printf("  | no-B-stage: %6.1f", t);
      t = time_tile<3, 1, 8, 32, 32, 16, 4, 1, 2, 1, 1, 10>(a, d); printf("  | no-A-stage: %6.1f", t);
    }
  } else {
    float t;
    t = time_tile<3, 2, 8, 16, 64, 8, 2, 2, 2, 1, MATH>(a, d);   printf("  s2 128x64/kc8 %6.1f", t);
    t = time_tile<3, 2, 8, 8, 64, 8, 2, 2, 1, 1, MATH>(a, d);    printf("  s2 64x64/kc8 %6.1f", t);
    t = time_tile<3, 2, 8, 8, 64, 16, 2, 2, 1, 1, MATH>(a, d);   printf("  s2 64x64/kc16 %6.1f", t);
    t = time_tile<3, 2, 8, 16, 32, 16, 4, 1, 1, 1, MATH>(a, d);  printf("  s2 128x32/kc16 %6.1f", t);
  }
  printf("  us\n");
}

int main() {
  Bufs b;
  const size_t big = 20ull * 256 * 256 * 96;
  hipMalloc(&b.s0, big * 4); hipMalloc(&b.s1, big * 4); hipMalloc(&b.out, big * 4);
  hipMalloc(&b.wp, 64u << 20); hipMalloc(&b.sc, 4096); hipMalloc(&b.sh, 4096);
  hipMemset(b.s0, 0, big * 4); hipMemset(b.s1, 0, big * 4); hipMemset(b.wp, 0, 64u << 20);
  std::vector<float> ones(1024, 1.f);
  hipMemcpy(b.sc, ones.data(), 4096, hipMemcpyHostToDevice);
  hipMemcpy(b.sh, ones.data(), 4096, hipMemcpyHostToDevice);
  // random-ish operand fill on the device is not needed for timing of fp32 MFMA, but
  // DVFS favours zeros: fill activations with a cheap pattern
  std::vector<float> pat(1 << 20);
  unsigned r = 1; for (auto& x : pat) { r = r * 1664525u + 1013904223u; x = ((r >> 8) / 8388608.0f - 1.0f); }
  for (size_t off = 0; off + pat.size() <= big; off += pat.size()) {
    hipMemcpy(b.s0 + off, pat.data(), pat.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b.s1 + off, pat.data(), pat.size() * 4, hipMemcpyHostToDevice);
  }
  for (size_t off = 0; off + pat.size() <= (64u << 20) / 4; off += pat.size())
    hipMemcpy(b.wp + off, pat.data(), pat.size() * 4, hipMemcpyHostToDevice);
#define BOTH(...) sweep<0>(__VA_ARGS__, b); sweep<1>(__VA_ARGS__, b);
  BOTH("conv8_2 20x256x256 32->32", 20, 256, 256, 32, 0, 0, 32, 1)
  BOTH("conv8_1 20x256x256 96->32 up+cat", 20, 256, 256, 64, 32, 1, 32, 1)
  BOTH("conv7_2 20x128x128 64->64", 20, 128, 128, 64, 0, 0, 64, 1)
  BOTH("conv7_1 20x128x128 192->64 up+cat", 20, 128, 128, 128, 64, 1, 64, 1)
  BOTH("conv6_2 20x64x64 128->128", 20, 64, 64, 128, 0, 0, 128, 1)
  BOTH("conv6_1 20x64x64 384->128 up+cat", 20, 64, 64, 256, 128, 1, 128, 1)
  BOTH("conv5_2 20x32x32 256->256", 20, 32, 32, 256, 0, 0, 256, 1)
  BOTH("conv5_1 20x32x32 768->256 up+cat", 20, 32, 32, 512, 256, 1, 256, 1)
  BOTH("conv4_2 20x16x16 512->512", 20, 16, 16, 512, 0, 0, 512, 1)
  BOTH("conv1_1 20x256x256 32->64 s2", 20, 256, 256, 32, 0, 0, 64, 2)
  BOTH("conv2_1 20x128x128 64->128 s2", 20, 128, 128, 64, 0, 0, 128, 2)
  BOTH("conv3_1 20x64x64 128->256 s2", 20, 64, 64, 128, 0, 0, 256, 2)
  BOTH("conv4_1 20x32x32 256->512 s2", 20, 32, 32, 256, 0, 0, 512, 2)
  return 0;
}
