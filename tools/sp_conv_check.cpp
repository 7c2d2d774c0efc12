// Correctness + timing of the split-planar conv engine (dn_spconv2d*) against the
// already parity-tested NHWC engine (dn_conv2d, math = 1: identical arithmetic up to the
// fp32 summation order) on the layer shapes of the BASELINE workload.  Not part of the
// product library; links libdisconet_hip.so:
//   hipcc -O2 -std=c++17 -I include tools/sp_conv_check.cpp -L disconet_amd -ldisconet_hip \
//         -Wl,-rpath,'$ORIGIN/../disconet_amd' -o tools/sp_conv_check.bin
//   tools/sp_conv_check.bin [n_images] [quick]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "disconet_hip.h"

#define CK(x) do { int rc_ = (x); if (rc_ != 0) { printf("FAILED %s -> %d: %s\n", #x, rc_, dn_last_error()); exit(2); } } while (0)
#define HCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP %s: %s\n", #x, hipGetErrorString(e_)); exit(3); } } while (0)

// ---- guard bands (SPCHECK_GUARD=1): every device buffer of the run sits between two 256 KiB bands of a byte pattern that
// are verified after EVERY case -- an out-of-bounds WRITE of any launch (epilogue stores, K-slice partials, weight packing)
// lands in a band and is reported with the buffer and the offset.  (Out-of-bounds READS cannot fault on this path: every
// operand access is a buffer load against a descriptor of the exact image / weight bytes and returns 0 past num_records.)
static const size_t kGuard = 256 * 1024;
static const unsigned char kPat = 0xA5;
static bool g_guard = false;
struct GuardedAlloc { unsigned char* base; size_t bytes; };
static std::vector<GuardedAlloc> g_allocs;
static hipError_t gmalloc(void** p, size_t bytes) {
  if (!g_guard) return hipMalloc(p, bytes);
  const size_t padded = (bytes + 255) / 256 * 256;          // (the band behind starts at the next 256-byte boundary)
  unsigned char* base = nullptr;
  hipError_t e = hipMalloc(&base, padded + 2 * kGuard);
  if (e != hipSuccess) return e;
  e = hipMemset(base, kPat, padded + 2 * kGuard);
  g_allocs.push_back({base, bytes});
  *p = base + kGuard;
  return e;
}
static int g_guard_fail = 0;
static void check_guards(const char* after) {
  if (!g_guard) return;
  hipDeviceSynchronize();
  std::vector<unsigned char> h(kGuard + 256);
  for (size_t i = 0; i < g_allocs.size(); ++i) {
    const auto& a = g_allocs[i];
    for (int side = 0; side < 2; ++side) {
      const size_t off = side ? kGuard + a.bytes : 0, len = side ? kGuard + ((a.bytes + 255) / 256 * 256 - a.bytes) : kGuard;
      if (hipMemcpy(h.data(), a.base + off, len, hipMemcpyDeviceToHost) != hipSuccess) { printf("guard read failed\n"); exit(3); }
      for (size_t k = 0; k < len; ++k)
        if (h[k] != kPat) {
          printf("   GUARD BAND OVERWRITTEN after [%s]: buffer #%zu (%zu bytes), %s band, byte %zu = 0x%02x\n", after, i, a.bytes,
                 side ? "rear" : "front", k, h[k]);
          ++g_guard_fail;
          hipMemset(a.base + off, kPat, len);
          break;
        }
    }
  }
}
static hipError_t gfree(void* p) {
  if (!g_guard || p == nullptr) return hipFree(p);
  for (size_t i = 0; i < g_allocs.size(); ++i)
    if (g_allocs[i].base + kGuard == (unsigned char*)p) {
      GuardedAlloc a = g_allocs[i];
      g_allocs.erase(g_allocs.begin() + i);
      return hipFree(a.base);
    }
  printf("   gfree: %p is not a guarded buffer\n", p);
  return hipErrorInvalidValue;
}
#define hipMalloc(pp, n) gmalloc((void**)(pp), (n))
#define hipFree(p) (check_guards(__func__), gfree(p))

static unsigned g_seed = 12345;
static float frand() { g_seed = g_seed * 1664525u + 1013904223u; return ((g_seed >> 8) / 8388608.0f) - 1.0f; }

static float* dev_random(size_t n, float amp, bool nonneg = false) {
  std::vector<float> h(n);
  for (auto& x : h) { x = frand() * amp; if (nonneg) x = fabsf(x); }
  float* d; HCK(hipMalloc(&d, n * 4)); HCK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}

struct Timer {
  hipEvent_t e0, e1;
  Timer() { hipEventCreate(&e0); hipEventCreate(&e1); }
  template <class F> float us(F f, int iters = 10) {
    for (int i = 0; i < 2; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters * 1e3f;
  }
};

static double compare(const float* d_a, const float* d_b, size_t n, const char* what, double* maxref = nullptr) {
  std::vector<float> a(n), b(n);
  HCK(hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost));
  HCK(hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost));
  double md = 0, mr = 0; size_t bad = 0, first = n;
  for (size_t i = 0; i < n; ++i) {
    if (std::isnan(a[i]) || std::isnan(b[i])) { ++bad; if (first == n) first = i; continue; }
    const double d = fabs((double)a[i] - b[i]);
    if (d > md) { md = d; if (d > 1e-3) first = first == n ? i : first; }
    mr = fmax(mr, fabs((double)b[i]));
  }
  if (maxref) *maxref = mr;
  if (bad) printf("   [%s] %zu NaNs (first at %zu)\n", what, bad, first);
  return bad ? 1e30 : md / (mr > 0 ? mr : 1);
}

static int g_fail = 0;

// where a failing result differs: error counts by (y & 7, x & 1) class and by channel octet, first few positions
static void dump_mismatch(const float* d_a, const float* d_b, int n, int h, int w, int c) {
  const size_t tot = (size_t)n * h * w * c;
  std::vector<float> a(tot), b(tot);
  hipMemcpy(a.data(), d_a, tot * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d_b, tot * 4, hipMemcpyDeviceToHost);
  double mr = 0;
  for (size_t i = 0; i < tot; ++i) mr = fmax(mr, fabs((double)b[i]));
  long by_row[8] = {0}, by_colpar[2] = {0}, by_x32[32] = {0}, by_oct[64] = {0}, bad = 0;
  int shown = 0;
  printf("\n   mismatch map (tolerance 2e-5 of max |ref| = %.3g):", mr);
  for (size_t i = 0; i < tot; ++i) {
    const bool nan = std::isnan(a[i]);
    if (!nan && fabs((double)a[i] - b[i]) <= 2e-5 * mr) continue;
    const int ch = i % c; size_t r = i / c;
    const int x = r % w; r /= w;
    const int y = r % h; const int im = (int)(r / h);
    ++bad; ++by_row[y & 7]; ++by_colpar[x & 1]; ++by_x32[x & 31]; ++by_oct[(ch / 8) & 63];
    if (shown < 6) { printf("\n     img %d y %d x %d ch %d: got %g want %g", im, y, x, ch, a[i], b[i]); ++shown; }
  }
  printf("\n     %ld of %zu bad; by y&7:", bad, tot);
  for (int k = 0; k < 8; ++k) printf(" %ld", by_row[k]);
  printf("; by x&1: %ld %ld; by x&31:", by_colpar[0], by_colpar[1]);
  for (int k = 0; k < 32; ++k) printf(" %ld", by_x32[k]);
  printf("; by channel octet:");
  for (int k = 0; k < (c + 7) / 8 && k < 64; ++k) printf(" %ld", by_oct[k]);
  printf("\n");
}

// names of conv_sp.hip's tile menu
static const char* kCfgName[] = {"256x64", "256x32", "128x64", "64x64", "s2_128x64", "s2_64x64", "p256x64", "p64x64",
                                 "256x64/T9", "512x64", "256x128", "p256x64/C1", "256x32/stat", "64x64/T9",
                                 "128x64/T9", "s2_64x64/T9", "s2_128x64/T9"};
static const char* g_filter = nullptr;   // substring of the layer name; several separated by ','
static bool name_selected(const char* name) {
  if (!g_filter) return true;
  const char* p = g_filter;
  while (*p) {
    const char* e = strchr(p, ',');
    const size_t len = e ? (size_t)(e - p) : strlen(p);
    if (len && std::string(name).find(std::string(p, len)) != std::string::npos) return true;
    p += len + (e ? 1 : 0);
  }
  return false;
}
static std::vector<int> g_only_cfgs;     // SPCHECK_CFGS=0,8,9: only these forced tile configurations (and no "auto")
static int g_mode = 0;                   // 0 = every tile, 1 = automatic selection only, 2 = + ablations

static void run_layer(const char* name, int n, int h, int w, int c0, int c1, int up0, int cout, int ks,
                      int stride, bool quick) {
  if (!name_selected(name)) return;
  Timer tm;
  dn_conv_desc d = {n, h, w, c0, c1, up0, cout, ks, stride, 1, c0, c1, cout, 1};
  const int ho = (h + 2 * (ks / 2) - ks) / stride + 1, wo = (w + 2 * (ks / 2) - ks) / stride + 1;
  const int h0 = up0 ? h / 2 : h, w0 = up0 ? w / 2 : w;
  const size_t n0 = (size_t)n * h0 * w0 * c0, n1 = (size_t)n * h * w * c1, no = (size_t)n * ho * wo * cout;
  float* s0 = dev_random(n0, 1.5f, true);
  float* s1 = c1 ? dev_random(n1, 1.5f, true) : nullptr;
  const int cin = c0 + c1;
  const float wamp = sqrtf(6.f / (cin * ks * ks));
  float* wt = dev_random((size_t)cout * cin * ks * ks, wamp);
  float* sc = dev_random(cout, 0.5f, true); float* sh = dev_random(cout, 0.3f);
  // scale in [0.75, 1.25]
  { std::vector<float> hsc(cout); for (auto& x : hsc) x = 1.f + 0.25f * frand(); hipMemcpy(sc, hsc.data(), cout * 4, hipMemcpyHostToDevice); }
  float *out_ref, *out_new;
  HCK(hipMalloc(&out_ref, no * 4)); HCK(hipMalloc(&out_new, no * 4));
  // ---- reference engine
  float* pk_ref; HCK(hipMalloc(&pk_ref, dn_conv_packed_weight_floats(&d) * 4));
  CK(dn_conv_pack_weights(&d, wt, pk_ref, 0));
  CK(dn_conv2d(&d, s0, s1, pk_ref, sc, sh, out_ref, 0));
  const float t_ref = tm.us([&] { dn_conv2d(&d, s0, s1, pk_ref, sc, sh, out_ref, 0); });
  // ---- SP engine; weights pre-scaled by 2^8 with 2^-8 folded into the scale
  void *sp0, *sp1 = nullptr, *spo, *pk = nullptr;
  HCK(hipMalloc(&sp0, dn_sp_tensor_bytes(n, h0, w0, c0)));
  if (c1) HCK(hipMalloc(&sp1, dn_sp_tensor_bytes(n, h, w, c1)));
  HCK(hipMalloc(&spo, dn_sp_tensor_bytes(n, ho, wo, cout)));
  HCK(hipMemset(spo, 0xFF, dn_sp_tensor_bytes(n, ho, wo, cout)));   // NaN poison
  CK(dn_sp_from_nhwc(s0, n, h0, w0, c0, c0, sp0, 0));
  if (c1) CK(dn_sp_from_nhwc(s1, n, h, w, c1, c1, sp1, 0));
  const float wmul = 256.f;
  float* sc2; HCK(hipMalloc(&sc2, cout * 4));
  { std::vector<float> hsc(cout); hipMemcpy(hsc.data(), sc, cout * 4, hipMemcpyDeviceToHost); for (auto& x : hsc) x /= wmul; hipMemcpy(sc2, hsc.data(), cout * 4, hipMemcpyHostToDevice); }
  const double gf = 2.0 * n * ho * wo * cout * (double)cin * ks * ks / 1e9;
  printf("%-30s %7.2f GF  ref %7.1f us (%6.1f TF) |", name, gf, t_ref, gf / t_ref * 1e3);
  // layers over an upsampled source: the row-merged image / tiles (mode 1), then the quad-merged kernel (mode 2:
  // automatic BN, then BN = 32 and BN = 64 forced); everything else: one pass
  for (int upmode = up0 ? 1 : 2; upmode <= 2; ++upmode) {
  dn_spconv_set_upmode(upmode);
  if (pk) hipFree(pk);
  HCK(hipMalloc(&pk, dn_spconv_packed_weight_bytes(&d)));
  CK(dn_spconv_pack_weights(&d, wt, wmul, pk, 0));
  std::vector<int> cfgs = {-1};
  if (up0 && upmode == 2) {
    printf(" [quad]");
    cfgs.insert(cfgs.end(), {20, 21, 22});
    if (g_mode == 2) cfgs.insert(cfgs.end(), {23, 24, 25});   // timing only: no weight DMA / no patch DMA / neither
  } else if (!quick) {
    if (ks == 1) cfgs.insert(cfgs.end(), {6, 7});
    else if (stride == 2) cfgs.insert(cfgs.end(), {4, 5, 15, 16});
    else cfgs.insert(cfgs.end(), {0, 1, 2, 3, 8, 9, 10, 12, 13, 14});
    if (g_mode == 2 && ks == 3 && stride == 2 && getenv("SPCHECK_S2_ABL")) cfgs.insert(cfgs.end(), {400, 401, 402, 403, 404, 405, 407});   // a -DDN_S2_ABL=1 library
    if (g_mode == 2 && ks == 3 && stride == 1) cfgs.insert(cfgs.end(), {101, 102, 103, 104, 105, 201, 202, 203, 204, 205, 206, 207, 301, 302, 303, 304, 305});
  }
  if (!g_only_cfgs.empty()) {
    std::vector<int> keep;
    for (int c : cfgs) for (int o : g_only_cfgs) if (c == o) keep.push_back(c);
    cfgs = keep;
  }
  for (int cfg : cfgs) {
    dn_spconv_force_config(cfg);
    if (cfg >= 100 || (cfg >= 23 && cfg <= 25)) {   // ablation: timing only (results are garbage by construction)
      if (cfg >= 200 && cout > 32 && 0) continue;
      CK(dn_spconv2d(&d, sp0, sp1, pk, sc2, sh, spo, 0));
      const float t = tm.us([&] { dn_spconv2d(&d, sp0, sp1, pk, sc2, sh, spo, 0); });
      printf(" abl%d %6.1f us |", cfg, t);
      continue;
    }
    HCK(hipMemset(spo, 0xFF, dn_sp_tensor_bytes(n, ho, wo, cout)));
    if (dn_spconv2d(&d, sp0, sp1, pk, sc2, sh, spo, 0) != 0) {   // a forced tile that does not apply
      printf(" %s n/a |", cfg < 0 ? "auto" : cfg >= 20 ? "q" : kCfgName[cfg]);
      continue;
    }
    CK(dn_sp_to_nhwc(spo, n, ho, wo, cout, cout, out_new, 0));
    HCK(hipDeviceSynchronize());
    const double err = compare(out_new, out_ref, no, name);
    const float t = tm.us([&] { dn_spconv2d(&d, sp0, sp1, pk, sc2, sh, spo, 0); });
    const bool ok = err < 2e-5;
    if (!ok) ++g_fail;
    printf(" %s %6.1f us %6.1f TF err %.1e%s |", cfg < 0 ? "auto" : cfg == 20 ? "q32" : cfg == 21 ? "q64" : cfg == 22 ? "q32/deep" : kCfgName[cfg], t,
           gf / t * 1e3, err, ok ? "" : " FAIL");
    if (!ok) dump_mismatch(out_new, out_ref, n, ho, wo, cout);
  }
  }
  dn_spconv_set_upmode(-1);
  dn_spconv_force_config(-1);
  printf("\n");
  fflush(stdout);
  hipFree(s0); hipFree(s1); hipFree(wt); hipFree(sc); hipFree(sh); hipFree(out_ref); hipFree(out_new);
  hipFree(pk_ref); hipFree(sp0); hipFree(sp1); hipFree(spo); hipFree(pk); hipFree(sc2);
}

// fused 3x3 (cin -> 64) + 1x1 (64 -> c2): fp32 two-output form (heads) and SP form (conv1_2 + Conv3D)
static void run_post(const char* name, int n, int h, int w, int cin, int c2, int split, bool f32, bool block = false) {
  if (!name_selected(name)) return;
  Timer tm;
  dn_conv_desc d = {n, h, w, cin, 0, 0, 64, 3, 1, 1, cin, 0, 64, 1};
  dn_post1x1_desc p = {c2, f32 ? 0 : 1, split, split, c2 - split, block ? 1 : 0};
  const size_t px = (size_t)n * h * w;
  float* s0 = dev_random(px * cin, 1.5f, true);
  float* wt = dev_random((size_t)64 * cin * 9, sqrtf(6.f / (cin * 9)));
  float* w2 = dev_random((size_t)c2 * 64, sqrtf(6.f / 64));
  if (block) {   // two heads side by side: rows < split read hidden 0..31, the rest 32..63
    std::vector<float> hw2((size_t)c2 * 64);
    hipMemcpy(hw2.data(), w2, hw2.size() * 4, hipMemcpyDeviceToHost);
    for (int r = 0; r < c2; ++r)
      for (int k = 0; k < 64; ++k)
        if ((r < split) != (k < 32)) hw2[(size_t)r * 64 + k] = 0.f;
    hipMemcpy(w2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice);
  }
  float* sc = dev_random(64, 0.5f, true); float* sh = dev_random(64, 0.3f);
  float* sc2 = dev_random(64, 0.5f, true); float* sh2 = dev_random(64, 0.3f);
  { std::vector<float> v(64); for (auto& x : v) x = 1.f + 0.25f * frand(); hipMemcpy(sc, v.data(), 256, hipMemcpyHostToDevice);
    for (auto& x : v) x = 1.f + 0.25f * frand(); hipMemcpy(sc2, v.data(), 256, hipMemcpyHostToDevice); }
  float *ra, *rb = nullptr, *na, *nb = nullptr;
  HCK(hipMalloc(&ra, px * split * 4)); HCK(hipMalloc(&na, px * split * 4));
  if (split < c2) { HCK(hipMalloc(&rb, px * (c2 - split) * 4)); HCK(hipMalloc(&nb, px * (c2 - split) * 4)); }
  float *pk_ref, *pk2_ref;
  HCK(hipMalloc(&pk_ref, dn_conv_packed_weight_floats(&d) * 4)); HCK(hipMalloc(&pk2_ref, dn_post1x1_packed_floats() * 4));
  CK(dn_conv_pack_weights(&d, wt, pk_ref, 0)); CK(dn_post1x1_pack_weights(w2, c2, 64, pk2_ref, 0));
  CK(dn_conv2d_post1x1(&d, &p, s0, nullptr, pk_ref, sc, sh, pk2_ref, sc2, sh2, ra, rb, 0));
  const float t_ref = tm.us([&] { dn_conv2d_post1x1(&d, &p, s0, nullptr, pk_ref, sc, sh, pk2_ref, sc2, sh2, ra, rb, 0); });
  void *sp0, *pk, *pk2, *spo = nullptr;
  HCK(hipMalloc(&sp0, dn_sp_tensor_bytes(n, h, w, cin))); CK(dn_sp_from_nhwc(s0, n, h, w, cin, cin, sp0, 0));
  HCK(hipMalloc(&pk, dn_spconv_packed_weight_bytes(&d))); HCK(hipMalloc(&pk2, dn_sp_post1x1_packed_bytes()));
  const float wmul = 64.f;
  CK(dn_spconv_pack_weights(&d, wt, wmul, pk, 0));
  if (block) CK(dn_sp_post1x1_pack_heads(w2, c2, split, wmul, pk2, 0));
  else CK(dn_sp_post1x1_pack_weights(w2, c2, 64, wmul, pk2, 0));
  float *scs, *sc2s; HCK(hipMalloc(&scs, 256)); HCK(hipMalloc(&sc2s, 256));
  { std::vector<float> v(64); hipMemcpy(v.data(), sc, 256, hipMemcpyDeviceToHost); for (auto& x : v) x /= wmul; hipMemcpy(scs, v.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(v.data(), sc2, 256, hipMemcpyDeviceToHost); for (auto& x : v) x /= wmul; hipMemcpy(sc2s, v.data(), 256, hipMemcpyHostToDevice); }
  double err;
  float t;
  if (f32) {
    HCK(hipMemset(na, 0xFF, px * split * 4)); if (nb) HCK(hipMemset(nb, 0xFF, px * (c2 - split) * 4));
    CK(dn_spconv2d_post1x1(&d, &p, sp0, nullptr, pk, scs, sh, pk2, sc2s, sh2, 1, na, nb, 0));
    HCK(hipDeviceSynchronize());
    err = compare(na, ra, px * split, name);
    if (nb) err = fmax(err, compare(nb, rb, px * (c2 - split), name));
    t = tm.us([&] { dn_spconv2d_post1x1(&d, &p, sp0, nullptr, pk, scs, sh, pk2, sc2s, sh2, 1, na, nb, 0); });
  } else {
    HCK(hipMalloc(&spo, dn_sp_tensor_bytes(n, h, w, c2))); HCK(hipMemset(spo, 0xFF, dn_sp_tensor_bytes(n, h, w, c2)));
    CK(dn_spconv2d_post1x1(&d, &p, sp0, nullptr, pk, scs, sh, pk2, sc2s, sh2, 0, spo, nullptr, 0));
    CK(dn_sp_to_nhwc(spo, n, h, w, c2, c2, na, 0));
    HCK(hipDeviceSynchronize());
    err = compare(na, ra, px * c2, name);
    t = tm.us([&] { dn_spconv2d_post1x1(&d, &p, sp0, nullptr, pk, scs, sh, pk2, sc2s, sh2, 0, spo, nullptr, 0); });
  }
  const double gf = 2.0 * px * (64.0 * cin * 9 + (double)c2 * 64) / 1e9;
  const bool ok = err < 2e-5;
  if (!ok) ++g_fail;
  printf("%-30s %7.2f GF  ref %7.1f us (%6.1f TF) | sp %6.1f us %6.1f TF err %.1e%s\n", name, gf, t_ref,
         gf / t_ref * 1e3, t, gf / t * 1e3, err, ok ? "" : " FAIL");
  fflush(stdout);
  hipFree(s0); hipFree(wt); hipFree(w2); hipFree(sc); hipFree(sh); hipFree(sc2); hipFree(sh2); hipFree(ra); hipFree(rb);
  hipFree(na); hipFree(nb); hipFree(pk_ref); hipFree(pk2_ref); hipFree(sp0); hipFree(pk); hipFree(pk2); hipFree(spo);
  hipFree(scs); hipFree(sc2s);
}

// conv_pre_1: 13 -> 32 over a 0/1 occupancy grid, full SP source vs the hi-only form (math = 3): bit-equal, timed
static void run_hi_only(const char* name, int n, int h, int w, int cin, int cout) {
  if (!name_selected(name)) return;
  Timer tm;
  dn_conv_desc d = {n, h, w, cin, 0, 0, cout, 3, 1, 1, cin, 0, cout, 2};
  const size_t nx = (size_t)n * h * w * cin, no = (size_t)n * h * w * cout;
  std::vector<float> hx(nx);
  for (auto& x : hx) x = frand() > 0.9f ? 1.f : 0.f;
  float* x; HCK(hipMalloc(&x, nx * 4)); HCK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice));
  float* wt = dev_random((size_t)cout * cin * 9, sqrtf(6.f / (cin * 9)));
  float* sc = dev_random(cout, 0.01f, true); float* sh = dev_random(cout, 0.3f);
  const size_t full_b = dn_sp_tensor_bytes(n, h, w, cin), plane = (size_t)h * w * 16;
  const int cg = (cin + 15) / 16;
  void *spf, *sph, *pk, *o1, *o2;
  HCK(hipMalloc(&spf, full_b)); HCK(hipMalloc(&sph, full_b / 2));
  CK(dn_sp_from_nhwc(x, n, h, w, cin, cin, spf, 0));
  for (int i = 0; i < n * cg; ++i)   // quarters 0, 1 (the hi octets) of every chunk
    HCK(hipMemcpy((char*)sph + (size_t)i * 2 * plane, (char*)spf + (size_t)i * 4 * plane, 2 * plane, hipMemcpyDeviceToDevice));
  HCK(hipMalloc(&pk, dn_spconv_packed_weight_bytes(&d)));
  CK(dn_spconv_pack_weights(&d, wt, 256.f, pk, 0));
  const size_t ob = dn_sp_tensor_bytes(n, h, w, cout);
  HCK(hipMalloc(&o1, ob)); HCK(hipMalloc(&o2, ob)); HCK(hipMemset(o1, 0xFF, ob)); HCK(hipMemset(o2, 0xFF, ob));
  CK(dn_spconv2d(&d, spf, nullptr, pk, sc, sh, o1, 0));
  const float t_full = tm.us([&] { dn_spconv2d(&d, spf, nullptr, pk, sc, sh, o1, 0); });
  dn_conv_desc dh = d; dh.math = 3;
  CK(dn_spconv2d(&dh, sph, nullptr, pk, sc, sh, o2, 0));
  const float t_hi = tm.us([&] { dn_spconv2d(&dh, sph, nullptr, pk, sc, sh, o2, 0); });
  float *f1, *f2; HCK(hipMalloc(&f1, no * 4)); HCK(hipMalloc(&f2, no * 4));
  CK(dn_sp_to_nhwc(o1, n, h, w, cout, cout, f1, 0)); CK(dn_sp_to_nhwc(o2, n, h, w, cout, cout, f2, 0));
  HCK(hipDeviceSynchronize());
  const double err = compare(f2, f1, no, name);
  const bool ok = err == 0.0;
  if (!ok) { ++g_fail; dump_mismatch(f2, f1, n, h, w, cout); }
  printf("%-30s full %6.1f us | hi-only %6.1f us | max rel diff %.1e%s\n", name, t_full, t_hi, err, ok ? "" : " FAIL");
  hipFree(x); hipFree(wt); hipFree(sc); hipFree(sh); hipFree(spf); hipFree(sph); hipFree(pk); hipFree(o1); hipFree(o2);
  hipFree(f1); hipFree(f2);
}

// K-sliced form (dn_spconv2d_ks): value against the reference engine, and the property it is built for -- the outputs
// of an image do not depend on the launch: n images with the tail split through the workspace == the same launch with
// nothing split (workspace NULL) == the first 4 images as a launch of their own, BIT FOR BIT.  Timed beside dn_spconv2d.
static void run_ks(const char* name, int n, int h, int w, int c0, int c1, int up0, int cout, int stride, int S) {
  if (!name_selected(name)) return;
  Timer tm;
  const int ks = 3;
  dn_conv_desc d = {n, h, w, c0, c1, up0, cout, ks, stride, 1, c0, c1, cout, 1};
  const int ho = (h + 2 - ks) / stride + 1, wo = (w + 2 - ks) / stride + 1;
  const int h0 = up0 ? h / 2 : h, w0 = up0 ? w / 2 : w;
  const size_t n0 = (size_t)n * h0 * w0 * c0, n1 = (size_t)n * h * w * c1, no = (size_t)n * ho * wo * cout;
  float* s0 = dev_random(n0, 1.5f, true);
  float* s1 = c1 ? dev_random(n1, 1.5f, true) : nullptr;
  const int cin = c0 + c1;
  float* wt = dev_random((size_t)cout * cin * 9, sqrtf(6.f / (cin * 9)));
  float* sc = dev_random(cout, 0.5f, true); float* sh = dev_random(cout, 0.3f);
  const float wmul = 256.f;
  { std::vector<float> hsc(cout); for (auto& x : hsc) x = (1.f + 0.25f * frand()); hipMemcpy(sc, hsc.data(), cout * 4, hipMemcpyHostToDevice); }
  float *out_ref, *out_new; HCK(hipMalloc(&out_ref, no * 4)); HCK(hipMalloc(&out_new, no * 4));
  float* pk_ref; HCK(hipMalloc(&pk_ref, dn_conv_packed_weight_floats(&d) * 4));
  CK(dn_conv_pack_weights(&d, wt, pk_ref, 0));
  CK(dn_conv2d(&d, s0, s1, pk_ref, sc, sh, out_ref, 0));
  float* sc2; HCK(hipMalloc(&sc2, cout * 4));
  { std::vector<float> hsc(cout); hipMemcpy(hsc.data(), sc, cout * 4, hipMemcpyDeviceToHost); for (auto& x : hsc) x /= wmul; hipMemcpy(sc2, hsc.data(), cout * 4, hipMemcpyHostToDevice); }
  void *sp0, *sp1 = nullptr, *o_full, *o_nows, *o_small, *pk, *ws;
  const size_t ob = dn_sp_tensor_bytes(n, ho, wo, cout), ob4 = dn_sp_tensor_bytes(4, ho, wo, cout);
  HCK(hipMalloc(&sp0, dn_sp_tensor_bytes(n, h0, w0, c0)));
  if (c1) HCK(hipMalloc(&sp1, dn_sp_tensor_bytes(n, h, w, c1)));
  HCK(hipMalloc(&o_full, ob)); HCK(hipMalloc(&o_nows, ob)); HCK(hipMalloc(&o_small, ob4));
  HCK(hipMemset(o_full, 0xFF, ob)); HCK(hipMemset(o_nows, 0xFF, ob)); HCK(hipMemset(o_small, 0xFF, ob4));
  CK(dn_sp_from_nhwc(s0, n, h0, w0, c0, c0, sp0, 0));
  if (c1) CK(dn_sp_from_nhwc(s1, n, h, w, c1, c1, sp1, 0));
  HCK(hipMalloc(&pk, dn_spconv_packed_weight_bytes(&d)));
  CK(dn_spconv_pack_weights(&d, wt, wmul, pk, 0));
  const size_t wsb = dn_spconv_workspace_bytes(&d, S);
  HCK(hipMalloc(&ws, wsb ? wsb : 16));
  HCK(hipMemset(ws, 0xFF, wsb ? wsb : 16));
  dn_conv_desc d4 = d; d4.n_images = n < 4 ? n : 4;
  CK(dn_spconv2d_ks(&d, S, sp0, sp1, pk, sc2, sh, o_full, nullptr, 0, ws, wsb, 0));
  CK(dn_spconv2d_ks(&d, S, sp0, sp1, pk, sc2, sh, o_nows, nullptr, 0, nullptr, 0, 0));
  CK(dn_spconv2d_ks(&d4, S, sp0, sp1, pk, sc2, sh, o_small, nullptr, 0, ws, wsb, 0));      // SP tensors are image-major: a prefix
  CK(dn_sp_to_nhwc(o_full, n, ho, wo, cout, cout, out_new, 0));
  HCK(hipDeviceSynchronize());
  const double err = compare(out_new, out_ref, no, name);
  std::vector<unsigned char> a(ob), b(ob), c(ob4);
  HCK(hipMemcpy(a.data(), o_full, ob, hipMemcpyDeviceToHost)); HCK(hipMemcpy(b.data(), o_nows, ob, hipMemcpyDeviceToHost));
  HCK(hipMemcpy(c.data(), o_small, ob4, hipMemcpyDeviceToHost));
  const size_t per_img = ob / n;
  const bool same_nows = memcmp(a.data(), b.data(), ob) == 0;
  const bool same_small = memcmp(a.data(), c.data(), per_img * d4.n_images) == 0;
  const float t_ks = tm.us([&] { dn_spconv2d_ks(&d, S, sp0, sp1, pk, sc2, sh, o_full, nullptr, 0, ws, wsb, 0); });
  const float t_nows = tm.us([&] { dn_spconv2d_ks(&d, S, sp0, sp1, pk, sc2, sh, o_full, nullptr, 0, nullptr, 0, 0); });
  const float t_plain = tm.us([&] { dn_spconv2d(&d, sp0, sp1, pk, sc2, sh, o_full, 0); });
  const float t_ks4 = tm.us([&] { dn_spconv2d_ks(&d4, S, sp0, sp1, pk, sc2, sh, o_small, nullptr, 0, ws, wsb, 0); });
  const float t_plain4 = tm.us([&] { dn_spconv2d(&d4, sp0, sp1, pk, sc2, sh, o_small, 0); });
  const bool ok = err < 2e-5 && same_nows && same_small;
  if (!ok) ++g_fail;
  printf("[ks%d] %-30s err %.1e | split == unsplit %s | 4-image launch == rows of the %d-image one %s | %d img: plain %6.1f  sliced/unsplit %6.1f  sliced %6.1f us | 4 img: plain %6.1f  sliced %6.1f us%s\n",
         S, name, err, same_nows ? "bitwise" : "DIFFERS", n, same_small ? "bitwise" : "DIFFERS", n, t_plain, t_nows, t_ks, t_plain4, t_ks4,
         ok ? "" : " FAIL");
  if (err >= 2e-5) dump_mismatch(out_new, out_ref, n, ho, wo, cout);
  fflush(stdout);
  hipFree(s0); hipFree(s1); hipFree(wt); hipFree(sc); hipFree(sh); hipFree(out_ref); hipFree(out_new); hipFree(pk_ref);
  hipFree(sp0); hipFree(sp1); hipFree(o_full); hipFree(o_nows); hipFree(o_small); hipFree(pk); hipFree(ws); hipFree(sc2);
}

static void roundtrip() {
  const int n = 2, h = 20, w = 24, c = 45;
  float* s = dev_random((size_t)n * h * w * c, 3.f);
  void* sp; HCK(hipMalloc(&sp, dn_sp_tensor_bytes(n, h, w, c)));
  float* back; HCK(hipMalloc(&back, (size_t)n * h * w * c * 4));
  CK(dn_sp_from_nhwc(s, n, h, w, c, c, sp, 0)); CK(dn_sp_to_nhwc(sp, n, h, w, c, c, back, 0));
  HCK(hipDeviceSynchronize());
  const double err = compare(back, s, (size_t)n * h * w * c, "roundtrip");
  printf("nhwc -> sp -> nhwc round trip (45 ch): rel err %.2e %s\n", err, err < 1e-6 ? "ok" : "FAIL");
  if (!(err < 1e-6)) ++g_fail;
}

int main(int argc, char** argv) {
  // sp_conv_check.bin [n_images] [layer-name filter | all] [tiles | auto | abl]
  const int n = argc > 1 ? atoi(argv[1]) : 20;
  g_guard = getenv("SPCHECK_GUARD") && atoi(getenv("SPCHECK_GUARD"));
  if (argc > 2 && strcmp(argv[2], "all")) g_filter = argv[2];
  if (argc > 3) g_mode = !strcmp(argv[3], "auto") ? 1 : !strcmp(argv[3], "abl") ? 2 : 0;
  const bool quick = g_mode == 1;
  if (const char* e = getenv("SPCHECK_CFGS")) for (const char* p = e; *p; ) { g_only_cfgs.push_back(atoi(p)); p = strchr(p, ','); if (!p) break; ++p; }
  printf("dn_version %d, %d images\n", dn_version(), n);
  roundtrip();
  // odd shapes first: ragged maps, channel counts that are not tile multiples
  run_layer("ragged 3x3 40x72 48->80", 2, 40, 72, 48, 0, 0, 80, 3, 1, quick);
  run_layer("ragged 3x3 s2 40x72 32->96", 2, 40, 72, 32, 0, 0, 96, 3, 2, quick);
  run_layer("ragged 3x3 up+cat 24x40", 2, 24, 40, 32, 16, 1, 48, 3, 1, quick);
  run_layer("ragged 3x3 up+cat 10x70 c1=24", 3, 10, 70, 16, 24, 1, 40, 3, 1, quick);
  run_layer("ragged 3x3 up only 16x32", 2, 16, 32, 32, 0, 1, 96, 3, 1, quick);
  run_layer("ragged 1x1 24x40 64->96", 2, 24, 40, 64, 0, 0, 96, 1, 1, quick);
  run_post("post f32 2x40x64 32->64->48", 2, 40, 64, 32, 48, 12, true);
  run_post("post f32 blockdiag 3x40x72", 3, 40, 72, 32, 48, 12, true, true);
  run_post("post f32 blockdiag 96->64->8", 2, 24, 40, 96, 8, 4, true, true);
  run_post("post sp  2x40x64 64->64->64", 2, 40, 64, 64, 64, 64, false);
  run_hi_only("hi-only 3x40x72 13->32", 3, 40, 72, 13, 32);
  run_hi_only("hi-only 2x24x40 20->80", 2, 24, 40, 20, 80);
  // the BASELINE layers
  run_hi_only("conv_pre_1 256^2 13->32", n, 256, 256, 13, 32);
  run_layer("conv_pre_2 256^2 32->32", n, 256, 256, 32, 0, 0, 32, 3, 1, quick);
  run_layer("conv1_1 256^2 32->64 s2", n, 256, 256, 32, 0, 0, 64, 3, 2, quick);
  run_layer("conv1_2 128^2 64->64", n, 128, 128, 64, 0, 0, 64, 3, 1, quick);
  run_layer("conv2_1 128^2 64->128 s2", n, 128, 128, 64, 0, 0, 128, 3, 2, quick);
  run_layer("conv2_2 64^2 128->128", n, 64, 64, 128, 0, 0, 128, 3, 1, quick);
  run_layer("conv3d_2 64^2 128->128 1x1", n, 64, 64, 128, 0, 0, 128, 1, 1, quick);
  run_layer("conv3_1 64^2 128->256 s2", n, 64, 64, 128, 0, 0, 256, 3, 2, quick);
  run_layer("conv3_2 32^2 256->256", n, 32, 32, 256, 0, 0, 256, 3, 1, quick);
  run_layer("conv4_1 32^2 256->512 s2", n, 32, 32, 256, 0, 0, 512, 3, 2, quick);
  run_layer("conv4_2 16^2 512->512", n, 16, 16, 512, 0, 0, 512, 3, 1, quick);
  run_layer("conv5_1 32^2 768->256 up+cat", n, 32, 32, 512, 256, 1, 256, 3, 1, quick);
  run_layer("conv6_1 64^2 384->128 up+cat", n, 64, 64, 256, 128, 1, 128, 3, 1, quick);
  run_layer("conv7_1 128^2 192->64 up+cat", n, 128, 128, 128, 64, 1, 64, 3, 1, quick);
  run_layer("conv8_1 256^2 96->32 up+cat", n, 256, 256, 64, 32, 1, 32, 3, 1, quick);
  run_layer("conv8_2 256^2 32->32", n, 256, 256, 32, 0, 0, 32, 3, 1, quick);
  run_layer("mlp 1x1 32^2 256->256", n, 32, 32, 256, 0, 0, 256, 1, 1, quick);
  run_post("heads 256^2 32->64->48 f32", n, 256, 256, 32, 48, 12, true);
  run_post("heads 256^2 blockdiag f32", n, 256, 256, 32, 48, 12, true, true);
  run_post("conv1_2+3d 128^2 64->64->64", n, 128, 128, 64, 64, 64, false);
  // one rank's share of the agent-sharded step (4 images): the deep layers under-fill the chip
  run_layer("share conv3_1 64^2 128->256 s2", 4, 64, 64, 128, 0, 0, 256, 3, 2, quick);
  run_layer("share conv3_2 32^2 256->256", 4, 32, 32, 256, 0, 0, 256, 3, 1, quick);
  run_layer("share conv4_1 32^2 256->512 s2", 4, 32, 32, 256, 0, 0, 512, 3, 2, quick);
  run_layer("share conv4_2 16^2 512->512", 4, 16, 16, 512, 0, 0, 512, 3, 1, quick);
  run_layer("share conv5_1 32^2 768->256 up+cat", 4, 32, 32, 512, 256, 1, 256, 3, 1, quick);
  run_layer("share conv5_2 32^2 256->256", 4, 32, 32, 256, 0, 0, 256, 3, 1, quick);
  run_layer("share conv6_1 64^2 384->128 up+cat", 4, 64, 64, 256, 128, 1, 128, 3, 1, quick);
  run_layer("share conv6_2 64^2 128->128", 4, 64, 64, 128, 0, 0, 128, 3, 1, quick);
  // K-sliced form of the deep layers (model.py :: _KSLICES), odd shapes first
  if (g_mode != 2) {
    run_ks("ks ragged 3x3 40x72 64->80", 6, 40, 72, 64, 0, 0, 80, 1, 4);
    run_ks("ks ragged 3x3 s2 40x72 64->96", 6, 40, 72, 64, 0, 0, 96, 2, 2);
    run_ks("ks ragged up+cat 24x40", 6, 24, 40, 48, 32, 1, 48, 1, 4);
    run_ks("ks ragged up only 16x32", 5, 16, 32, 64, 0, 1, 96, 1, 2);
    run_ks("ks conv3_2 32^2 256->256", n, 32, 32, 256, 0, 0, 256, 1, 4);
    run_ks("ks conv4_1 32^2 256->512 s2", n, 32, 32, 256, 0, 0, 512, 2, 4);
    run_ks("ks conv4_2 16^2 512->512", n, 16, 16, 512, 0, 0, 512, 1, 4);
    run_ks("ks conv5_1 32^2 768->256 up+cat", n, 32, 32, 512, 256, 1, 256, 1, 4);
    run_ks("ks conv5_2 32^2 256->256", n, 32, 32, 256, 0, 0, 256, 1, 4);
    run_ks("ks conv6_1 64^2 384->128 up+cat", n, 64, 64, 256, 128, 1, 128, 1, 4);
    run_ks("ks conv6_2 64^2 128->128", n, 64, 64, 128, 0, 0, 128, 1, 2);
    run_ks("ks conv2_2 64^2 128->128 (4)", n, 64, 64, 128, 0, 0, 128, 1, 4);
    run_ks("ks conv3_1 64^2 128->256 s2", n, 64, 64, 128, 0, 0, 256, 2, 2);
  }
  check_guards("end of run");
  if (g_guard) printf("guard bands: %zu buffers, %d overwritten\n", g_allocs.size(), g_guard_fail);
  g_fail += g_guard_fail;
  printf("%s (%d failures)\n", g_fail ? "SP CONV CHECK FAILED" : "SP CONV CHECK PASSED", g_fail);
  return g_fail ? 1 : 0;
}
