// conv_pre_1 -> conv_pre_2 of the encoder stem in ONE launch (included by conv_sp.hip: same translation unit, same range flags).
//
//   upstream:coperception/models/det/backbone/Backbone.py :: Backbone.encode, the first two layers
//   (conv_pre_1 13 -> 32, conv_pre_2 32 -> 32, both 3x3 / BN / ReLU at the full 256 x 256 map).
//
// Why: as two launches the pair moves conv_pre_1's 168 MB output through HBM twice (written, then read back 35 us later) and
// both launches run near the memory roofline (4.7 / 4.2 TB/s, profiles/r04_pmc_conv_sp.txt) with the MFMA pipe a third busy.
// Fused, the intermediate never leaves the CU: a workgroup (8 waves, one per CU) owns a 16 x 32 output tile,
//   stage 1  computes conv_pre_1 on the tile's 18 x 34 halo patch straight from the occupancy words (20 MFMA pixel tiles
//            of 32, a word's byte -> its f16 0 / 1 fragment through a 256-entry table in LDS: AHI = 2's arithmetic), applies affine + ReLU +
//            hi/lo split and writes the SP pieces into LDS in the patch layout conv_pre_2's K loop reads -- zero where the
//            patch pixel lies outside the map (that is conv_pre_2's zero padding, not relu(bias));
//   stage 2  is conv_pre_2's weight-stationary K loop over that patch (both 16-channel chunks resident: no DMA, no stage
//            hand-over) and conv_sp_kernel's register epilogue.
// Every accumulation chain has the order of the two-launch path (stage 1: per tap w_lo.x then w_hi.x, taps 0..8; stage 2:
// per tap w_lo.x_hi, w_hi.x_lo, w_hi.x_hi, chunks 0, 1) and the same affine4 / split4, so the output is bit-identical to
// dn_spconv2d(math 4) followed by dn_spconv2d; 19.5 % of stage 1 is halo recompute.
// A pipelined form (8 x 32 tiles, two patch buffers, four producer waves running stage 1 of the next tile beside four consumer
// waves on stage 2, one barrier per tile; in the repository's history as "Stem pair, pipelined form") measured the same 2342 vs 2347 scenes/s in one lease, as did
// conv_pre_1's weight fragments held in registers: the launch sits at the busy x clock plateau of the engine's other layers
// (DESIGN.md 3.1e), not on its phase structure.  The simpler two-phase kernel is the one kept.
// LDS: patch 2 x 39168 + conv_pre_2 weights 36864 + conv_pre_1 weights 18432 + occupancy words 2880 + affines 512 + the
// byte -> fragment table 4096 = 141120 B.

#ifndef DN_PRE_LUT
#define DN_PRE_LUT 1     // tools/ab: 0 = the occupancy bytes expanded with VALU instructions instead of a table in LDS
#endif

namespace {

namespace pp {
constexpr int TH = 16, TW = 32;                       // output tile
constexpr int MH = TH + 2, MW = TW + 2, MNPIX = MH * MW;          // intermediate patch: 18 x 34 = 612 pixels
constexpr int MTILES = (MNPIX + 31) / 32;             // 20 MFMA pixel tiles (the last holds 4 pixels)
constexpr int BH = TH + 4, BW = TW + 4, BWORDS = BH * BW;         // occupancy words: 20 x 36 = 720
constexpr int NWAVE = 8, NTHR = NWAVE * 64;
constexpr int MID_CHUNK = 4 * MNPIX * 16;             // one 16-channel chunk of the patch: [4 quarters][612] x 16 B
constexpr int W2_BYTES = 2 * 9 * 4 * 32 * 16, W1_BYTES = 9 * 4 * 32 * 16;
constexpr int OFF_W2 = 2 * MID_CHUNK, OFF_W1 = OFF_W2 + W2_BYTES, OFF_BITS = OFF_W1 + W1_BYTES, OFF_AFF = OFF_BITS + BWORDS * 4;
constexpr int OFF_LUT = OFF_AFF + 4 * 32 * 4;          // 256 x 16 B: occupancy byte -> its 8 halves
constexpr int LDS_BYTES = OFF_LUT + 256 * 16;
static_assert(OFF_BITS % 16 == 0 && OFF_AFF % 16 == 0 && LDS_BYTES <= 160 * 1024, "LDS layout");
constexpr int K1 = (MTILES + NWAVE - 1) / NWAVE;      // stage-1 pixel tiles per wave (3; waves 4..7 run 2)
}  // namespace pp

struct PrePairArgs {
  const unsigned* bits;          // [n][h][w] occupancy words
  const unsigned char* w1;       // packed conv_pre_1 weights (one chunk): [tap 9][quarter 4][cout_pad] x 16 B
  const unsigned char* w2;       // packed conv_pre_2 weights: [chunk 2][tap 9][quarter 4][cout_pad] x 16 B
  int cout_pad;
  const float *s1, *t1, *s2, *t2;
  unsigned char* out;            // SP tensor [n][cog][4][h][w] x 16 B
  int n, h, w, c_out, cog, relu1, relu2;
  int tiles_x, tiles_y, items;
  float rcp_tx, rcp_ty;
};

// byte m of an occupancy word -> the 8 halves (0x3C00 where the bit is set) of the hi-only stage: conv_sp_kernel's commit_a
__device__ inline half8 expand_octet(unsigned m) {
  u32x4 v;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned x = (m >> (2 * k)) & 3u;                    // bits 2k, 2k + 1
    v[k] = (((x << 15) | x) & 0x10001u) * 0x3C00u;             // bit 2k -> low half, bit 2k + 1 -> high half
  }
  return __builtin_bit_cast(half8, v);
}

__global__ void __launch_bounds__(pp::NTHR, 1) conv_pre_pair_kernel(const PrePairArgs a) {
  using namespace pp;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int G = gridDim.x;
  int item = blockIdx.x;
  if (item >= a.items) return;

  // ---- once per workgroup: both layers' weights and affines -> LDS
  // (packed images: [chunk][tap][4 quarters][cout_pad] pieces -- the first 32 channels of every quarter row)
  auto piece_of = [&](int i) { return (size_t)(((i >> 7) * 4 + ((i >> 5) & 3)) * a.cout_pad + (i & 31)) * 16; };
  for (int i = tid; i < W2_BYTES / 16; i += NTHR)
    *reinterpret_cast<u32x4*>(smem + OFF_W2 + i * 16) = *reinterpret_cast<const u32x4*>(a.w2 + piece_of(i));
  for (int i = tid; i < W1_BYTES / 16; i += NTHR)
    *reinterpret_cast<u32x4*>(smem + OFF_W1 + i * 16) = *reinterpret_cast<const u32x4*>(a.w1 + piece_of(i));
  float* aff = reinterpret_cast<float*>(smem + OFF_AFF);       // s1, t1, s2, t2 (channels past c_out: 0 -> the outputs are 0)
  if (tid < 32) {
    aff[tid] = a.s1[tid];
    aff[32 + tid] = a.t1[tid];
    aff[64 + tid] = tid < a.c_out ? a.s2[tid] : 0.f;
    aff[96 + tid] = tid < a.c_out ? a.t2[tid] : 0.f;
  }

#if DN_PRE_LUT
  if (tid < 256) *reinterpret_cast<half8*>(smem + OFF_LUT + tid * 16) = expand_octet((unsigned)tid);
#endif
#if DN_PHASE_TIMING
  unsigned long long ph[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = __builtin_readcyclecounter();
  const unsigned long long t_begin = t_prev;
  auto mark = [&](int k) { const unsigned long long t = __builtin_readcyclecounter(); ph[k] += t - t_prev; t_prev = t; };
#else
  auto mark = [&](int) {};
#endif
  const int sp_full = a.items & ~7;
  auto decode = [&](int it) {
    TileCoord tc;
    int spi = it < sp_full ? (it & 7) * (sp_full >> 3) + (it >> 3) : it;     // XCD-aware order, as conv_sp_kernel
    int tx, ty;
    spi = fdivmod(spi, a.tiles_x, a.rcp_tx, tx);
    tc.img = fdivmod(spi, a.tiles_y, a.rcp_ty, ty);
    tc.img = __builtin_amdgcn_readfirstlane(tc.img);
    tc.ox0 = __builtin_amdgcn_readfirstlane(tx * TW);
    tc.oy0 = __builtin_amdgcn_readfirstlane(ty * TH);
    tc.n0 = 0;
    return tc;
  };

  // ---- occupancy words of a tile: word idx of the 20 x 36 block = tid, tid + 512 (< 720); out of the map: 0
  const size_t img_words = (size_t)a.h * a.w;
  int bw_r[2], bw_c[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + q * NTHR;
    bw_r[q] = idx / BW;
    bw_c[q] = idx % BW;
  }
  unsigned wreg[2];
  auto load_words = [&](const TileCoord& tc) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(a.bits + (size_t)tc.img * img_words), 0,
                                                        (int)(img_words * 4), 0x00020000);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int iy = tc.oy0 - 2 + bw_r[q], ix = tc.ox0 - 2 + bw_c[q];
      const bool ok = tid + q * NTHR < BWORDS && iy >= 0 && iy < a.h && ix >= 0 && ix < a.w;
      wreg[q] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, ok ? (unsigned)(iy * a.w + ix) * 4u : 0xFFFFFFFFu, 0, 0);
    }
  };
  auto store_words = [&]() {
    unsigned* bl = reinterpret_cast<unsigned*>(smem + OFF_BITS);
    bl[tid] = wreg[0];
    if (tid + NTHR < BWORDS) bl[tid + NTHR] = wreg[1];
  };

  // ---- stage-1 constants of this lane: patch pixel pos = 32 t + li of pixel tile t = wave + 8 k
  int s1_word[K1], s1_r[K1], s1_c[K1], s1_dst[K1];
  bool s1_valid[K1];
#pragma unroll
  for (int k = 0; k < K1; ++k) {
    const int pos = 32 * (wave + NWAVE * k) + li;
    s1_valid[k] = pos < MNPIX;
    const int pc = s1_valid[k] ? pos : MNPIX - 1;
    s1_r[k] = pc / MW;
    s1_c[k] = pc % MW;
    s1_word[k] = (s1_r[k] * BW + s1_c[k]) * 4;                  // byte offset of tap (0, 0)'s word in the block
    s1_dst[k] = (lh * MNPIX + pc) * 16;                          // hi piece of octet lh; lo: + 2 MNPIX * 16; chunk 1: + MID_CHUNK
  }
  const int b_off = (lh * 32 + li) * 16;                         // this lane's weight fragment inside a tap block [4][32] x 16 B
  // ---- stage-2 constants: output rows 2 wave, 2 wave + 1 of the tile, column li
  int a_off[2];
#pragma unroll
  for (int wm = 0; wm < 2; ++wm) a_off[wm] = (lh * MNPIX + (2 * wave + wm) * MW + li) * 16;

  float amax = 0.f;
  bool nan_seen = false;
  TileCoord cur = decode(item);
  load_words(cur);
  store_words();
  __syncthreads();
  // both stages' affines of this lane's channels 8 g + 4 lh + e: registers for the whole launch
  f32x4 sc1[4], sh1[4], sc2[4], sh2[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    sc1[g] = *reinterpret_cast<const f32x4*>(aff + 8 * g + 4 * lh);
    sh1[g] = *reinterpret_cast<const f32x4*>(aff + 32 + 8 * g + 4 * lh);
    sc2[g] = *reinterpret_cast<const f32x4*>(aff + 64 + 8 * g + 4 * lh);
    sh2[g] = *reinterpret_cast<const f32x4*>(aff + 96 + 8 * g + 4 * lh);
  }
  const float lo_clamp1 = a.relu1 ? 0.f : -65504.f, lo_clamp2 = a.relu2 ? 0.f : -65504.f;
  const int plane = a.h * a.w * 16, img_bytes = a.cog * 4 * plane;

  while (true) {
    // ================= stage 1: conv_pre_1 on the halo patch -> LDS =================
#pragma unroll
    for (int k = 0; k < K1; ++k) {
      if (wave + NWAVE * k >= MTILES) break;                     // wave-uniform
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const unsigned char* wp = smem + OFF_BITS + s1_word[k];
      const unsigned char* w1p = smem + OFF_W1 + b_off;
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        const unsigned word = *reinterpret_cast<const unsigned*>(wp + ((u / 3) * BW + u % 3) * 4);
#if DN_PRE_LUT
        // (a table read instead of 17 VALU instructions; at LiDAR occupancies most lanes read entry 0: one broadcast)
        const half8 x = *reinterpret_cast<const half8*>(smem + OFF_LUT + ((word >> (8 * lh)) & 0xffu) * 16);
#else
        const half8 x = expand_octet((word >> (8 * lh)) & 0xffu);
#endif
        const half8 bh = *reinterpret_cast<const half8*>(w1p + u * 2048);
        const half8 bl = *reinterpret_cast<const half8*>(w1p + u * 2048 + 1024);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, x, acc, 0, 0, 0);
      }
      // affine + ReLU + split; a patch pixel outside the map is conv_pre_2's zero padding
      const int iy = cur.oy0 - 1 + s1_r[k], ix = cur.ox0 - 1 + s1_c[k];
      const bool in_map = iy >= 0 && iy < a.h && ix >= 0 && ix < a.w;
      u32x2 hi[4], lo[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = affine4(quad_of(acc, g), sc1[g], sh1[g]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = in_map ? v[e] : 0.f;
        split4(v, hi[g], lo[g], amax, lo_clamp1);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {                              // chunk m = channels 16 m .. 16 m + 15
        const u32x4 ph = gather_octet(hi[2 * m], hi[2 * m + 1]);
        const u32x4 pl = gather_octet(lo[2 * m], lo[2 * m + 1]);
        if (s1_valid[k]) {
          *reinterpret_cast<u32x4*>(smem + m * MID_CHUNK + s1_dst[k]) = ph;
          *reinterpret_cast<u32x4*>(smem + m * MID_CHUNK + s1_dst[k] + 2 * MNPIX * 16) = pl;
        }
      }
    }
    mark(0);
    __syncthreads();                                             // the patch is complete; the occupancy block is free
    mark(1);
    const bool has_next = item + G < a.items;
    TileCoord nxt = cur;
    if (has_next) {
      nxt = decode(item + G);
      load_words(nxt);                                           // lands under stage 2
    }
    // ================= stage 2: conv_pre_2 over the patch =================
    f32x16 acc2[2];
#pragma unroll
    for (int wm = 0; wm < 2; ++wm)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[wm][r] = 0.f;
    {
      half8 ah[2][2], al[2][2], bh[2], bl[2];
      auto load = [&](int s, int g, int u) {
        const unsigned char* As = smem + g * MID_CHUNK + ((u / 3) * MW + u % 3) * 16;
        const unsigned char* Bs = smem + OFF_W2 + (g * 9 + u) * 2048 + b_off;
#pragma unroll
        for (int wm = 0; wm < 2; ++wm) {
          ah[s][wm] = *reinterpret_cast<const half8*>(As + a_off[wm]);
          al[s][wm] = *reinterpret_cast<const half8*>(As + a_off[wm] + 2 * MNPIX * 16);
        }
        bh[s] = *reinterpret_cast<const half8*>(Bs);
        bl[s] = *reinterpret_cast<const half8*>(Bs + 1024);
      };
      auto mma = [&](int s) {
#pragma unroll
        for (int wm = 0; wm < 2; ++wm) acc2[wm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[s], ah[s][wm], acc2[wm], 0, 0, 0);
#pragma unroll
        for (int wm = 0; wm < 2; ++wm) acc2[wm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s], al[s][wm], acc2[wm], 0, 0, 0);
#pragma unroll
        for (int wm = 0; wm < 2; ++wm) acc2[wm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[s], ah[s][wm], acc2[wm], 0, 0, 0);
      };
      load(0, 0, 0);
#pragma unroll
      for (int q = 0; q < 18; ++q) {                             // (chunk, tap) = (q / 9, q % 9): reads of q + 1 before the MFMAs of q
        if (q + 1 < 18) load((q + 1) & 1, (q + 1) / 9, (q + 1) % 9);
        __builtin_amdgcn_sched_barrier(0);
        mma(q & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#if DN_PHASE_TIMING
    { float probe = acc2[1][0]; asm volatile("v_mov_b32 %0, %0" : "+v"(probe)); }
#endif
    mark(2);
    // ---- epilogue: conv_sp_kernel's store_sp_tile (register affine, buffer stores against the output image)
    {
      const auto rsrc_o = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)cur.img * img_bytes, 0, img_bytes, 0x00020000);
#pragma unroll
      for (int wm = 0; wm < 2; ++wm) {
        const int oy = cur.oy0 + 2 * wave + wm, ox = cur.ox0 + li;
        const bool inside = oy < a.h && ox < a.w;
        const int voff = inside ? (oy * a.w + ox) * 16 + lh * plane : (int)0x80000000;
        u32x2 hi[4], lo[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) split4(affine4(quad_of(acc2[wm], g), sc2[g], sh2[g]), hi[g], lo[g], amax, lo_clamp2);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const u32x4 ph = gather_octet(hi[2 * m], hi[2 * m + 1]);
          const u32x4 pl = gather_octet(lo[2 * m], lo[2 * m + 1]);
          if (m < a.cog) {
            __builtin_amdgcn_raw_buffer_store_b128(ph, rsrc_o, voff + m * 4 * plane, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(pl, rsrc_o, voff + (m * 4 + 2) * plane, 0, 0);
          }
        }
      }
    }
    mark(3);
#if DN_PHASE_TIMING
    ph[5] += 1;
#endif
    if (!has_next) break;
    store_words();                                               // the next tile's occupancy block (free since the barrier above)
    __syncthreads();                                             // every wave is done reading the patch; the words are in place
    mark(4);
    item += G;
    cur = nxt;
  }
  note_range(amax, nan_seen);
#if DN_PHASE_TIMING
  ph[6] = __builtin_readcyclecounter() - t_begin;
  if (lane == 0 && (wave == 0 || wave == 7))      // wave 0 runs 3 stage-1 pixel tiles, wave 7 runs 2
    for (int k = 0; k < 7; ++k) atomicAdd(&g_phase_cycles[k], ph[k]);
#endif
}


}  // namespace

extern "C" int dn_spconv2d_pre_pair(const dn_conv_desc* d1, const dn_conv_desc* d2, const uint32_t* bits, const void* packed1,
                                    const float* scale1, const float* shift1, const void* packed2, const float* scale2,
                                    const float* shift2, void* out, void* stream) {
  DN_REQUIRE(d1 && d2 && bits && packed1 && scale1 && shift1 && packed2 && scale2 && shift2 && out, "spconv pre pair: null pointer");
  DN_REQUIRE(dn_spconv2d_pre_pair_supported(d1, d2), "spconv pre pair: needs two 3x3 stride-1 single-source layers on one map, "
             "c0 <= 16 -> 32 -> c_out <= 32 (got %d -> %d, %d -> %d)", d1->c0, d1->c_out, d2->c0, d2->c_out);
  PrePairArgs a;
  a.bits = bits;
  a.w1 = (const unsigned char*)packed1; a.w2 = (const unsigned char*)packed2;
  a.s1 = scale1; a.t1 = shift1; a.s2 = scale2; a.t2 = shift2;
  a.out = (unsigned char*)out;
  a.n = d1->n_images; a.h = d1->h_in; a.w = d1->w_in; a.c_out = d2->c_out; a.cog = (d2->c_out + 15) / 16;
  a.relu1 = d1->relu; a.relu2 = d2->relu;
  a.cout_pad = cout_pad_of(32);      // both layers: c_out <= 32 -> the same padded row (dn_spconv_pack_weights)
  a.tiles_x = (a.w + pp::TW - 1) / pp::TW; a.tiles_y = (a.h + pp::TH - 1) / pp::TH;
  a.items = a.n * a.tiles_x * a.tiles_y;
  a.rcp_tx = 1.f / a.tiles_x; a.rcp_ty = 1.f / a.tiles_y;
  DN_REQUIRE((size_t)a.cog * 4 * a.h * a.w * 16 < ((size_t)1 << 31) && (size_t)a.h * a.w * 4 < ((size_t)1 << 31) && a.items < (1 << 22),
             "spconv pre pair: map too large");
  static dn::PerDeviceFlag attr_flag;
  bool& attr_set = attr_flag.here();
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pre_pair_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, pp::LDS_BYTES);
    if (e != hipSuccess) return dn::fail(DN_ERR_LAUNCH, "spconv pre pair: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int grid = a.items < kCUs ? a.items : kCUs;
  hipLaunchKernelGGL(conv_pre_pair_kernel, dim3(grid), dim3(pp::NTHR), pp::LDS_BYTES, (hipStream_t)stream, a);
  return dn::check_launch("conv_pre_pair_kernel");
}

extern "C" int dn_spconv2d_pre_pair_supported(const dn_conv_desc* d1, const dn_conv_desc* d2) {
  if (!d1 || !d2) return 0;
  auto plain3 = [](const dn_conv_desc* d) { return d->ksize == 3 && d->stride == 1 && d->c1 == 0 && d->up0 == 0; };
  return plain3(d1) && plain3(d2) && d1->c0 >= 1 && d1->c0 <= 16 && d1->c_out == 32 && d2->c0 == 32 && d2->c_out >= 1 &&
         d2->c_out <= 32 && d1->n_images == d2->n_images && d1->n_images > 0 && d1->h_in == d2->h_in && d1->w_in == d2->w_in &&
         d1->h_in > 0 && d1->w_in > 0;
}
