// Weight gradient on the f16 MFMA with split operands: 3x3 layers, 64 x 64 or 32 x 32 (co, ci) blocks per workgroup; the stride-1
// kernel first, the stride-2 kernel (column-parity planes of the patch) at the end of the file.  include/disconet_train.h ::
// dn_conv_wgrad_sp; replaces autograd's weight gradient of the nn.Conv2d layers (upstream:coperception/utils/CoDetModule.py :: step).
//
// Same GEMM as conv_wgrad64_kernel -- D[co][ci] += dz^T . x_shifted per tap, K = pixels -- but every value is an
// f16 hi + lo pair (x = hi + lo, 22 significand bits) and a product is three v_mfma_f32_32x32x16_f16
// (hi.hi + hi.lo + lo.hi, fp32 accumulate): 16 pixels per instruction at 16 x the fp32 MFMA's rate, 5.3 x per product.
// K = pixels means a lane's fragment is 8 CONSECUTIVE PIXELS of ONE channel, while the maps are NHWC: the transposition
// happens in the staging pass.  A thread loads the same 4 channels of two horizontally adjacent pixels (2 x 16 B),
// multiplies by the operand's power-of-two lift, splits, and writes one dword per channel = the PIXEL PAIR's two halves
// (ds_write_b128 per part): LDS image [pixel pair][channel] dwords.  A fragment is then 4 dword reads (pairs
// 4 kb .. 4 kb + 3 of the lane's channel); the x fragment of tap column 2 starts one pair later, and tap column 1 is
// v_alignbit of neighbouring dwords -- 5 reads per patch row and part serve the three tap columns.
// The dz tile ([4 | 8 rows][16 pixels]) and the x patch ([6 | 10][18]) of one pixel tile are prefetched into registers under the
// previous tile's MFMAs, as in the fp32 kernels; the per-slice partial blocks and their fixed-order sum are shared with them.
struct WgradSpArgs {
  const float* src0;
  const float* src1;
  const float* dz;
  float* partial;
  int n_images, h_in, w_in;
  int c0, c1, up0, c_out;
  int ld0, ld1, ldz;
  int tiles_x, tiles_y, n_tiles;
  int n_cot, n_cit, n_slices;
  int vecx;            // 0: the sources' rows are not 16-byte loadable (13 input channels): four dword loads, each with its own bound
  float dz_lift, x_lift;
  const unsigned char* dz_sp;      // ZSP kernels: dz * dz_lift as the SP tensor the BatchNorm backward wrote (include/disconet_hip.h "SP tensor")
};

// ZSP (round 6): dz comes PRE-SPLIT -- the SP copy of dz * lift that dn_bn_train_backward_finish_sp / _bias write for the data
// gradient, the same hi / lo halves this kernel's staging pass derives from the fp32 rows (same lift, same split: the results are
// the same bits) -- so the fp32 copy of dz need not be written at all, and the tile's staging is 16 byte permutes per 16 values
// instead of a multiply, two conversions and a subtraction per value.  A thread takes the hi and lo pieces (8 channels each) of
// two adjacent pixels and writes one dword per channel and part, as before.
__device__ inline void zsp_put(unsigned* H, int slot, const u32x4 a, const u32x4 b) {
  // a, b: 8 halves (channels 0..7) of pixel 0 / pixel 1 -> per channel one dword: pixel 0 in the low half, pixel 1 in the high half
  u32x4 lo4, hi4;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    lo4[2 * k] = __builtin_amdgcn_perm(b[k], a[k], 0x05040100u);
    lo4[2 * k + 1] = __builtin_amdgcn_perm(b[k], a[k], 0x07060302u);
    hi4[2 * k] = __builtin_amdgcn_perm(b[k + 2], a[k + 2], 0x05040100u);
    hi4[2 * k + 1] = __builtin_amdgcn_perm(b[k + 2], a[k + 2], 0x07060302u);
  }
  *reinterpret_cast<u32x4*>(&H[slot]) = lo4;
  *reinterpret_cast<u32x4*>(&H[slot + 4]) = hi4;
}

// CB = channels per side of a workgroup's (co, ci) block.
//   64: four waves = the four 32 x 32 quadrants, each over all pixels of a 4 x 16 tile (3x3 layers with >= 64 channels both sides).
//   32: ONE 32 x 32 block; the four waves split the rows of an 8 x 16 tile (two each) and meet in a fixed-order LDS sum at the end
//       (the 32-channel layers of the 256 x 256 maps).
template <int CB>
struct WspShape {
  static constexpr int TH = CB == 64 ? 4 : 8, TW = 16, PH = TH + 2;
  static constexpr int ROWS = CB == 64 ? TH : TH / 4;             // output rows of the tile one wave works on
  static constexpr int QN = CB / 4;                                // channel quads per pixel
  static constexpr int XPAIRS = PH * 9;                            // pixel pairs of the x patch (18 columns)
  static constexpr int DPAIRS = TH * 8;                            // ... of the dz tile
  static constexpr int PITCH = CB + 8;                             // dwords per pixel pair: lanes 32-63 read 4 pairs on -> other banks
  static constexpr int X_IT = (XPAIRS * QN + 255) / 256;           // staging rounds of the patch
  static constexpr int D_IT = DPAIRS * QN / 256;                   // ... of the dz tile (exact)
  static constexpr int LDS_DWORDS = 2 * (XPAIRS + DPAIRS) * PITCH; // >= 9 * 32 * 32 for the CB = 32 reduction
  static_assert(DPAIRS * QN % 256 == 0, "dz tile rounds");
  static_assert(CB == 64 || LDS_DWORDS >= 9 * 1024, "reduction buffer");
};

template <int CB, bool ZSP = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_sp_kernel(const WgradSpArgs a) {
  using S = WspShape<CB>;
  constexpr int QN8 = CB / 8, D_IT8 = (S::DPAIRS * QN8 + 255) / 256;      // ZSP: (pixel pair, channel octet) items of the dz tile
  constexpr int PITCH = S::PITCH, QN = S::QN, TW = S::TW, TH = S::TH;
  constexpr unsigned OOB = 0xFFFFFFFFu;
  extern __shared__ __attribute__((aligned(16))) unsigned wsp_smem[];
  unsigned* Xh = wsp_smem;
  unsigned* Xl = Xh + S::XPAIRS * PITCH;
  unsigned* Dh = Xl + S::XPAIRS * PITCH;
  unsigned* Dl = Dh + S::DPAIRS * PITCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = CB == 64 ? wave >> 1 : 0, wn = CB == 64 ? wave & 1 : 0;
  const int row0 = CB == 64 ? 0 : wave * S::ROWS;          // first output row of the tile this wave works on

  int item = blockIdx.x;
  const int slice = item % a.n_slices;
  item /= a.n_slices;
  const int cit = item % a.n_cit;
  const int cot = item / a.n_cit;
  const int co0 = cot * CB, ci0 = cit * CB;
  const bool from1 = ci0 >= a.c0;                          // (c0 is a multiple of CB: a block never straddles the sources)
  const int cs0 = from1 ? ci0 - a.c0 : ci0;
  const int csrc = from1 ? a.c1 : a.c0;
  const int ld = from1 ? a.ld1 : a.ld0;
  const bool up = !from1 && a.up0;
  const int hs = up ? a.h_in >> 1 : a.h_in, ws = up ? a.w_in >> 1 : a.w_in;
  const float* src = from1 ? a.src1 : a.src0;
  const size_t img_x = (size_t)hs * ws * ld, img_z = (size_t)a.h_in * a.w_in * a.ldz;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  f32x4 rx[S::X_IT][2], rd[ZSP ? 1 : S::D_IT][2];
  u32x4 rz[ZSP ? D_IT8 : 1][4];                       // ZSP: hi pieces of the two pixels, then their lo pieces
  float amax = 0.f;
  const size_t hw_z = (size_t)a.h_in * a.w_in;        // (stride 1: the output map is the input's size)

  auto ld128 = [](auto rsrc, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
  };
  auto ldu128 = [](auto rsrc, unsigned voff) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0)); };
  auto ld32x4 = [](auto rsrc, unsigned voff, int nvalid) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                           rsrc, (voff == 0xFFFFFFFFu || e >= nvalid) ? 0xFFFFFFFFu : voff + 4 * e, 0, 0));
    return v;
  };
  auto load_tile = [&](int tile) {
    int sp = tile;
    const int ox0 = (sp % a.tiles_x) * TW;
    sp /= a.tiles_x;
    const int oy0 = (sp % a.tiles_y) * TH;
    const int img = sp / a.tiles_y;
    const auto rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src + img * img_x), 0, (int)(img_x * 4), 0x00020000);
    const auto rsz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz + img * img_z), 0, (int)(img_z * 4), 0x00020000);
#pragma unroll
    for (int it = 0; it < S::X_IT; ++it) {
      const int idx = tid + it * 256;
      const int pr = idx / QN, q = idx % QN;
      const int prow = pr / 9, pp = pr - prow * 9;
      const int iy = oy0 - 1 + prow, c = cs0 + 4 * q;
      const bool rowok = idx < S::XPAIRS * QN && iy >= 0 && iy < a.h_in && c < csrc;
      const int sy = up ? iy >> 1 : iy;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ix = ox0 - 1 + 2 * pp + e;
        const bool ok = rowok && ix >= 0 && ix < a.w_in;
        const int sx = up ? ix >> 1 : ix;
        const unsigned off = ok ? (unsigned)(((sy * ws + sx) * ld + c) * 4) : OOB;
        rx[it][e] = a.vecx ? ld128(rsx, off) : ld32x4(rsx, off, csrc - c);
      }
    }
    if constexpr (ZSP) {
      // image = [c_out / 16][4 quarters][h][w] pieces of 16 bytes, quarter = 2 * part + octet (c_out % 16 == 0: as many bytes as the fp32 image)
      const auto rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.dz_sp + img * (hw_z * a.c_out * 4)), 0,
                                                         (int)(hw_z * a.c_out * 4), 0x00020000);
#pragma unroll
      for (int it = 0; it < D_IT8; ++it) {
        const int idx = tid + it * 256;
        const int pr = idx / QN8, o = idx % QN8;
        const int oy = oy0 + (pr >> 3), c = co0 + 8 * o;
        const bool rowok = idx < S::DPAIRS * QN8 && oy < a.h_in && c < a.c_out;
        const unsigned q0 = (unsigned)((c >> 4) * 4 + ((c >> 3) & 1));
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int ox = ox0 + 2 * (pr & 7) + e;
          const bool ok = rowok && ox < a.w_in;
          const unsigned px = (unsigned)(oy * a.w_in + ox);
          rz[it][e] = ldu128(rsp, ok ? (unsigned)((q0 * hw_z + px) * 16) : OOB);
          rz[it][2 + e] = ldu128(rsp, ok ? (unsigned)(((q0 + 2) * hw_z + px) * 16) : OOB);
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < S::D_IT; ++it) {
        const int idx = tid + it * 256;
        const int pr = idx / QN, q = idx % QN;
        const int oy = oy0 + (pr >> 3), c = co0 + 4 * q;
        const bool rowok = oy < a.h_in && c < a.c_out;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int ox = ox0 + 2 * (pr & 7) + e;
          rd[it][e] = ld128(rsz, (rowok && ox < a.w_in) ? (unsigned)(((oy * a.w_in + ox) * a.ldz + c) * 4) : OOB);
        }
      }
    }
  };
  // two pixels x four channels -> per channel one dword (pixel 0 in the low half, pixel 1 in the high half), hi and lo parts
  auto put = [&](unsigned* H, unsigned* L, int slot, const f32x4 p0, const f32x4 p1, float lift) {
    u32x2 h01, l01, h23, l23;
    split4(f32x4{p0[0] * lift, p1[0] * lift, p0[1] * lift, p1[1] * lift}, h01, l01, amax);
    split4(f32x4{p0[2] * lift, p1[2] * lift, p0[3] * lift, p1[3] * lift}, h23, l23, amax);
    *reinterpret_cast<u32x4*>(&H[slot]) = u32x4{h01[0], h01[1], h23[0], h23[1]};
    *reinterpret_cast<u32x4*>(&L[slot]) = u32x4{l01[0], l01[1], l23[0], l23[1]};
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int it = 0; it < S::X_IT; ++it) {
      const int idx = tid + it * 256;
      if (idx < S::XPAIRS * QN) put(Xh, Xl, (idx / QN) * PITCH + 4 * (idx % QN), rx[it][0], rx[it][1], a.x_lift);
    }
    if constexpr (ZSP) {
#pragma unroll
      for (int it = 0; it < D_IT8; ++it) {
        const int idx = tid + it * 256;
        if (idx < S::DPAIRS * QN8) {
          const int slot = (idx / QN8) * PITCH + 8 * (idx % QN8);
          zsp_put(Dh, slot, rz[it][0], rz[it][1]);
          zsp_put(Dl, slot, rz[it][2], rz[it][3]);
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < S::D_IT; ++it) {
        const int idx = tid + it * 256;
        put(Dh, Dl, (idx / QN) * PITCH + 4 * (idx % QN), rd[it][0], rd[it][1], a.dz_lift);
      }
    }
  };

  int tile = slice;
  if (tile < a.n_tiles) {
    load_tile(tile);
    store_tile();
  }
  __syncthreads();
  // this lane's x channel / dz channel at its pixel-pair group (lanes 32-63: pixels 8..15 of a row = 4 pairs on), from the
  // wave's first patch row / output row
  const int bcol = wn * 32 + li + (lh * 4 + row0 * 9) * PITCH;
  const int acol = wm * 32 + li + (lh * 4 + row0 * 8) * PITCH;
  for (; tile < a.n_tiles; tile += a.n_slices) {
    const bool more = tile + a.n_slices < a.n_tiles;
    if (more) load_tile(tile + a.n_slices);
#pragma unroll
    for (int prow = 0; prow < S::ROWS + 2; ++prow) {
      // one part of the patch row at a time (hi: products dz_hi.x_hi and dz_lo.x_hi; lo: dz_hi.x_lo): 9 fragment registers live
      // instead of 18 -- the accumulators (144) and the prefetched tile (48) leave no room for both parts
#pragma unroll
      for (int part = 0; part < 2; ++part) {
        const unsigned* X = part ? Xl : Xh;
        unsigned b[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) b[i] = X[(prow * 9 + i) * PITCH + bcol];
        half8 f[3];
        f[0] = __builtin_bit_cast(half8, u32x4{b[0], b[1], b[2], b[3]});
        f[1] = __builtin_bit_cast(half8, u32x4{__builtin_amdgcn_alignbit(b[1], b[0], 16), __builtin_amdgcn_alignbit(b[2], b[1], 16),
                                               __builtin_amdgcn_alignbit(b[3], b[2], 16), __builtin_amdgcn_alignbit(b[4], b[3], 16)});
        f[2] = __builtin_bit_cast(half8, u32x4{b[1], b[2], b[3], b[4]});
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          const int r = prow - ty;               // output row (of this wave's rows) that sees this patch row through tap row ty
          if (r < 0 || r >= S::ROWS) continue;
          u32x4 ah;
#pragma unroll
          for (int i = 0; i < 4; ++i) ah[i] = Dh[(r * 8 + i) * PITCH + acol];
          const half8 ahh = __builtin_bit_cast(half8, ah);
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) acc[ty * 3 + tx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahh, f[tx], acc[ty * 3 + tx], 0, 0, 0);
          if (part == 0) {
            u32x4 al;
#pragma unroll
            for (int i = 0; i < 4; ++i) al[i] = Dl[(r * 8 + i) * PITCH + acol];
            const half8 alh = __builtin_bit_cast(half8, al);
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) acc[ty * 3 + tx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alh, f[tx], acc[ty * 3 + tx], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
    if (more) store_tile();
    __syncthreads();
  }
  note_range(amax);

  float* out = a.partial + ((size_t)(slice * a.n_cot + cot) * a.n_cit + cit) * (9 * CB * CB);
  if constexpr (CB == 64) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = lh * 4 + (r & 3) + 8 * (r >> 2);
        out[t * 4096 + (wm * 32 + i) * 64 + wn * 32 + li] = acc[t][r];
      }
  } else {
    // the four waves hold the block for different rows of the tiles: add in wave order (fixed), then the block goes out
    float* R = reinterpret_cast<float*>(wsp_smem);
    for (int wv = 0; wv < 4; ++wv) {
      if (wave == wv) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = lh * 4 + (r & 3) + 8 * (r >> 2);
            float* slot = &R[t * 1024 + i * 32 + li];
            *slot = (wv == 0 ? 0.f : *slot) + acc[t][r];
          }
      }
      __syncthreads();
    }
    for (int i = tid * 4; i < 9 * 1024; i += 1024) *reinterpret_cast<f32x4*>(out + i) = *reinterpret_cast<const f32x4*>(&R[i]);
  }
}

// ---- stride 2 ------------------------------------------------------------------------------------------------------------
// dW[co][ci][ty][tx] = sum dz[oy][ox][co] * x[2 oy + ty - 1][2 ox + tx - 1][ci]: a fragment is still 8 consecutive OUTPUT pixels of a
// row, i.e. every second input column -- so a patch row is staged as two column-parity planes, O (input columns 2 (ox0 + j) - 1,
// j = 0..16: tap columns 0 and 2) and E (columns 2 (ox0 + j), j = 0..15: tap column 1), 9 + 8 pixel pairs per row.  Tap column 0
// reads O pairs 4 kb .. 4 kb + 3, tap column 1 the same E pairs, tap column 2 is O one entry on: v_alignbit of neighbouring dwords,
// as tap column 1 of the stride-1 kernel.  Output row r sees patch rows 2 r + ty.  The patch of a tile is ~4 x the stride-1 kernel's
// per output pixel, so the tiles are short: CB = 64: DN_WSP_S2_TH64 x 16 outputs (3 patch rows at 1), the four waves = the four quadrants;
// CB = 32: 4 x 16 outputs (9 patch rows), one output row per wave, fixed-order LDS sum at the end.  No upsample / concat (the
// stride-2 layers have one source).
#ifndef DN_WSP_S2_TH64
#define DN_WSP_S2_TH64 1          // output rows per tile of the CB = 64 stride-2 kernel; 2 (58 KB of LDS, 255 VGPRs) measured equal: profiles/r06_wgrad_s2_tile.txt
#endif
template <int CB>
struct WspShape2 {
  static constexpr int TH = CB == 64 ? DN_WSP_S2_TH64 : 4, TW = 16, PH = 2 * TH + 1;
  static constexpr int ROWS = CB == 64 ? TH : 1;                    // output rows each wave walks (CB = 32: one row per wave)
  static constexpr int QN = CB / 4;
  static constexpr int RP = 17;                                     // pixel pairs per patch row: 9 of plane O, 8 of plane E
  static constexpr int XPAIRS = PH * RP;
  static constexpr int DPAIRS = TH * 8;
  static constexpr int PITCH = CB + 8;
  static constexpr int X_IT = (XPAIRS * QN + 255) / 256;
  static constexpr int D_IT = (DPAIRS * QN + 255) / 256;
  static constexpr int LDS_DWORDS = 2 * (XPAIRS + DPAIRS) * PITCH;
  static_assert(CB == 64 || LDS_DWORDS >= 9 * 1024, "reduction buffer");
};

template <int CB, bool ZSP = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_sp_s2_kernel(const WgradSpArgs a) {
  using S = WspShape2<CB>;
  constexpr int QN8 = CB / 8, D_IT8 = (S::DPAIRS * QN8 + 255) / 256;      // ZSP: (pixel pair, channel octet) items of the dz tile
  constexpr int PITCH = S::PITCH, QN = S::QN, TW = S::TW, TH = S::TH, RP = S::RP;
  constexpr unsigned OOB = 0xFFFFFFFFu;
  extern __shared__ __attribute__((aligned(16))) unsigned wsp_smem[];
  unsigned* Xh = wsp_smem;
  unsigned* Xl = Xh + S::XPAIRS * PITCH;
  unsigned* Dh = Xl + S::XPAIRS * PITCH;
  unsigned* Dl = Dh + S::DPAIRS * PITCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = CB == 64 ? wave >> 1 : 0, wn = CB == 64 ? wave & 1 : 0;
  const int row0 = CB == 64 ? 0 : wave;                    // the output row of the tile this wave works on

  int item = blockIdx.x;
  const int slice = item % a.n_slices;
  item /= a.n_slices;
  const int cit = item % a.n_cit;
  const int cot = item / a.n_cit;
  const int co0 = cot * CB, ci0 = cit * CB;
  const int h_out = (a.h_in - 1) / 2 + 1, w_out = (a.w_in - 1) / 2 + 1;
  const size_t img_x = (size_t)a.h_in * a.w_in * a.ld0, img_z = (size_t)h_out * w_out * a.ldz;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  f32x4 rx[S::X_IT][2], rd[ZSP ? 1 : S::D_IT][2];
  u32x4 rz[ZSP ? D_IT8 : 1][4];                       // ZSP: hi pieces of the two pixels, then their lo pieces
  float amax = 0.f;
  const size_t hw_z = (size_t)h_out * w_out;

  auto ld128 = [](auto rsrc, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
  };
  auto ldu128 = [](auto rsrc, unsigned voff) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0)); };
  auto load_tile = [&](int tile) {
    int sp = tile;
    const int ox0 = (sp % a.tiles_x) * TW;
    sp /= a.tiles_x;
    const int oy0 = (sp % a.tiles_y) * TH;
    const int img = sp / a.tiles_y;
    const auto rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0 + img * img_x), 0, (int)(img_x * 4), 0x00020000);
    const auto rsz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz + img * img_z), 0, (int)(img_z * 4), 0x00020000);
#pragma unroll
    for (int it = 0; it < S::X_IT; ++it) {
      const int idx = tid + it * 256;
      const int pr = idx / QN, q = idx % QN;
      const int prow = pr / RP, pp = pr - prow * RP;
      const int iy = 2 * oy0 - 1 + prow, c = ci0 + 4 * q;
      const bool rowok = idx < S::XPAIRS * QN && iy >= 0 && iy < a.h_in && c < a.c0;
      // plane O (pp < 9): entries j = 2 pp, 2 pp + 1 at input column 2 (ox0 + j) - 1; plane E: j = 2 (pp - 9), + 1 at 2 (ox0 + j)
      const int ixb = pp < 9 ? 2 * (ox0 + 2 * pp) - 1 : 2 * (ox0 + 2 * (pp - 9));
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ix = ixb + 2 * e;
        const bool ok = rowok && ix >= 0 && ix < a.w_in;
        rx[it][e] = ld128(rsx, ok ? (unsigned)(((iy * a.w_in + ix) * a.ld0 + c) * 4) : OOB);
      }
    }
    if constexpr (ZSP) {
      const auto rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.dz_sp + img * (hw_z * a.c_out * 4)), 0,
                                                         (int)(hw_z * a.c_out * 4), 0x00020000);
#pragma unroll
      for (int it = 0; it < D_IT8; ++it) {
        const int idx = tid + it * 256;
        const int pr = idx / QN8, o = idx % QN8;
        const int oy = oy0 + (pr >> 3), c = co0 + 8 * o;
        const bool rowok = idx < S::DPAIRS * QN8 && oy < h_out && c < a.c_out;
        const unsigned q0 = (unsigned)((c >> 4) * 4 + ((c >> 3) & 1));
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int ox = ox0 + 2 * (pr & 7) + e;
          const bool ok = rowok && ox < w_out;
          const unsigned px = (unsigned)(oy * w_out + ox);
          rz[it][e] = ldu128(rsp, ok ? (unsigned)((q0 * hw_z + px) * 16) : OOB);
          rz[it][2 + e] = ldu128(rsp, ok ? (unsigned)(((q0 + 2) * hw_z + px) * 16) : OOB);
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < S::D_IT; ++it) {
        const int idx = tid + it * 256;
        const int pr = idx / QN, q = idx % QN;
        const int oy = oy0 + (pr >> 3), c = co0 + 4 * q;
        const bool rowok = idx < S::DPAIRS * QN && oy < h_out && c < a.c_out;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int ox = ox0 + 2 * (pr & 7) + e;
          rd[it][e] = ld128(rsz, (rowok && ox < w_out) ? (unsigned)(((oy * w_out + ox) * a.ldz + c) * 4) : OOB);
        }
      }
    }
  };
  auto put = [&](unsigned* H, unsigned* L, int slot, const f32x4 p0, const f32x4 p1, float lift) {
    u32x2 h01, l01, h23, l23;
    split4(f32x4{p0[0] * lift, p1[0] * lift, p0[1] * lift, p1[1] * lift}, h01, l01, amax);
    split4(f32x4{p0[2] * lift, p1[2] * lift, p0[3] * lift, p1[3] * lift}, h23, l23, amax);
    *reinterpret_cast<u32x4*>(&H[slot]) = u32x4{h01[0], h01[1], h23[0], h23[1]};
    *reinterpret_cast<u32x4*>(&L[slot]) = u32x4{l01[0], l01[1], l23[0], l23[1]};
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int it = 0; it < S::X_IT; ++it) {
      const int idx = tid + it * 256;
      if (idx < S::XPAIRS * QN) put(Xh, Xl, (idx / QN) * PITCH + 4 * (idx % QN), rx[it][0], rx[it][1], a.x_lift);
    }
    if constexpr (ZSP) {
#pragma unroll
      for (int it = 0; it < D_IT8; ++it) {
        const int idx = tid + it * 256;
        if (idx < S::DPAIRS * QN8) {
          const int slot = (idx / QN8) * PITCH + 8 * (idx % QN8);
          zsp_put(Dh, slot, rz[it][0], rz[it][1]);
          zsp_put(Dl, slot, rz[it][2], rz[it][3]);
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < S::D_IT; ++it) {
        const int idx = tid + it * 256;
        if (idx < S::DPAIRS * QN) put(Dh, Dl, (idx / QN) * PITCH + 4 * (idx % QN), rd[it][0], rd[it][1], a.dz_lift);
      }
    }
  };

  int tile = slice;
  if (tile < a.n_tiles) {
    load_tile(tile);
    store_tile();
  }
  __syncthreads();
  const int bcol = wn * 32 + li + (lh * 4 + 2 * row0 * RP) * PITCH;     // patch row 2 row0 (+ ty), this lane's pixel-pair group
  const int acol = wm * 32 + li + (lh * 4 + row0 * 8) * PITCH;
  for (; tile < a.n_tiles; tile += a.n_slices) {
    const bool more = tile + a.n_slices < a.n_tiles;
    if (more) load_tile(tile + a.n_slices);
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      const unsigned* X = part ? Xl : Xh;
#pragma unroll
      for (int r = 0; r < S::ROWS; ++r) {
        const int ac = acol + r * 8 * PITCH, bc = bcol + 2 * r * RP * PITCH;
        u32x4 ah;
#pragma unroll
        for (int i = 0; i < 4; ++i) ah[i] = Dh[i * PITCH + ac];
        const half8 ahh = __builtin_bit_cast(half8, ah);
        half8 alh = ahh;
        if (part == 0) {
          u32x4 al;
#pragma unroll
          for (int i = 0; i < 4; ++i) al[i] = Dl[i * PITCH + ac];
          alh = __builtin_bit_cast(half8, al);
        }
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          unsigned o[5], ev[4];
#pragma unroll
          for (int i = 0; i < 5; ++i) o[i] = X[(ty * RP + i) * PITCH + bc];
#pragma unroll
          for (int i = 0; i < 4; ++i) ev[i] = X[(ty * RP + 9 + i) * PITCH + bc];
          half8 f[3];
          f[0] = __builtin_bit_cast(half8, u32x4{o[0], o[1], o[2], o[3]});
          f[1] = __builtin_bit_cast(half8, u32x4{ev[0], ev[1], ev[2], ev[3]});
          f[2] = __builtin_bit_cast(half8, u32x4{__builtin_amdgcn_alignbit(o[1], o[0], 16), __builtin_amdgcn_alignbit(o[2], o[1], 16),
                                                 __builtin_amdgcn_alignbit(o[3], o[2], 16), __builtin_amdgcn_alignbit(o[4], o[3], 16)});
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) acc[ty * 3 + tx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahh, f[tx], acc[ty * 3 + tx], 0, 0, 0);
          if (part == 0) {
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) acc[ty * 3 + tx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alh, f[tx], acc[ty * 3 + tx], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
    if (more) store_tile();
    __syncthreads();
  }
  note_range(amax);

  float* out = a.partial + ((size_t)(slice * a.n_cot + cot) * a.n_cit + cit) * (9 * CB * CB);
  if constexpr (CB == 64) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = lh * 4 + (r & 3) + 8 * (r >> 2);
        out[t * 4096 + (wm * 32 + i) * 64 + wn * 32 + li] = acc[t][r];
      }
  } else {
    float* R = reinterpret_cast<float*>(wsp_smem);
    for (int wv = 0; wv < 4; ++wv) {
      if (wave == wv) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = lh * 4 + (r & 3) + 8 * (r >> 2);
            float* slot = &R[t * 1024 + i * 32 + li];
            *slot = (wv == 0 ? 0.f : *slot) + acc[t][r];
          }
      }
      __syncthreads();
    }
    for (int i = tid * 4; i < 9 * 1024; i += 1024) *reinterpret_cast<f32x4*>(out + i) = *reinterpret_cast<const f32x4*>(&R[i]);
  }
}
