// Index maps of the split-planar ("SP") activation format and of the LDS images
// the SP conv engine (conv_sp.hip) stages with LDS-DMA.  Pure integer functions,
// usable from host code (tests/c_abi/check_sp_layout.cpp checks bijection and
// ds_read_b128 bank-conflict freedom on the CPU) and from the kernels.
//
// SP activation tensor:  [image][C/16 chunk][4 quarter][H][W] x 16 bytes
//   one 16-byte piece = 8 halves = channels 16*chunk + 8*oct + 0..7 of one pixel,
//   quarter q = 2*part + oct, part 0 = hi = half(x), part 1 = lo = half(x - hi).
//   A value is x ~= hi + lo (22-bit significand); bytes per element = 4, as fp32.
//   This is the B-operand fragment order of v_mfma_f32_32x32x16_f16 (lane (j, h)
//   holds k = 8h..8h+7), so a plane row is at once an HBM burst, an LDS-DMA burst
//   and a conflict-free ds_read_b128 source.
//
// SP packed weights:     [chunk][tap][4 quarter][cout_pad] x 16 bytes
//   piece = 8 halves = input channels 16*chunk + 8*oct + 0..7 of output channel n.
#pragma once

#if defined(__HIPCC__)
#define SP_HD __host__ __device__ inline
#else
#define SP_HD inline
#endif

namespace sp {

// ds_read_b128 is serviced in four 16-lane groups; within the low 32 lanes they are
// g1 = {0-3, 12-15, 20-27} and g2 = {4-11, 16-19, 28-31} (MI355X_MICROARCH.md, LDS).
SP_HD constexpr bool in_g2(int j) { return (j >= 4 && j < 12) || (j >= 16 && j < 20) || j >= 28; }
SP_HD constexpr int rank16(int j) {   // position of lane j inside its 16-lane group
  return j < 4 ? j : j < 12 ? j - 4 : j < 16 ? j - 8 : j < 20 ? j - 8 : j < 28 ? j - 12 : j - 16;
}

// Pixel (row, col) inside the workgroup's TH x TW output tile that MFMA column j of the
// pixel group `gm` (32 pixels) owns.  Chosen so that every 16-lane read group touches 16
// distinct 16-byte LDS slots (mod 16) for every tap:
//   TW = 32: a group is one tile row, column = j (any 16 of 32 consecutive pieces differ mod 16);
//   TW = 16: a group is two rows; each 16-lane read group takes one whole row;
//   TW =  8: four rows {g, g + 4} x {g1, g2}: rows r and r + 4 are 4 * STRIDE * PITCH pieces
//            apart, = 8 (mod 16) for the natural pitches 10 (3x3 s1) and 17 (3x3 s2).
template <int TW>
SP_HD constexpr int tile_row(int gm, int j) {
  if (TW == 32) return gm;
  if (TW == 16) return gm * 2 + (in_g2(j) ? 1 : 0);
  return gm * 2 + (in_g2(j) ? 1 : 0) + 4 * (rank16(j) / 8);   // TW == 8
}
template <int TW>
SP_HD constexpr int tile_col(int j) {
  if (TW == 32) return j;
  if (TW == 16) return rank16(j);
  return rank16(j) % 8;
}

// patch geometry of one workgroup: input rows/cols an output tile reads
template <int KS, int STRIDE, int TH, int TW>
struct Patch {
  static constexpr int PH = (TH - 1) * STRIDE + KS;
  static constexpr int PW = (TW - 1) * STRIDE + KS;
  // stride 2: columns are stored de-interleaved, EVENW even columns then the odd ones, so a
  // tap reads consecutive pieces for consecutive output columns
  static constexpr int EVENW = STRIDE == 2 ? TW + 1 : 0;
  // pieces per patch row; 1x1 with TW = 8 pads 8 -> 10 to keep rows r, r + 4 apart by 8 (mod 16)
  static constexpr int PITCH = (KS == 1 && TW == 8) ? 10 : PW;
  static constexpr int NPIX = PH * PITCH;   // pieces of one (chunk, quarter) plane of the patch
};

// LDS position (piece index inside a quarter plane) of patch pixel (r, cc)
template <int KS, int STRIDE, int TH, int TW>
SP_HD constexpr int patch_pos(int r, int cc) {
  using P = Patch<KS, STRIDE, TH, TW>;
  return r * P::PITCH + (STRIDE == 2 ? ((cc & 1) ? P::EVENW + (cc >> 1) : (cc >> 1)) : cc);
}
// inverse: patch column of position `pos` within a row, or -1 for padding
template <int KS, int STRIDE, int TH, int TW>
SP_HD constexpr int patch_col_of(int pos) {
  using P = Patch<KS, STRIDE, TH, TW>;
  if (STRIDE == 2) {
    if (pos < P::EVENW) return 2 * pos;
    const int cc = 2 * (pos - P::EVENW) + 1;
    return cc < P::PW ? cc : -1;
  }
  return pos < P::PW ? pos : -1;
}
// piece offset a tap adds to an output pixel's base position
template <int KS, int STRIDE, int TH, int TW>
SP_HD constexpr int tap_offset(int ty, int tx) {
  using P = Patch<KS, STRIDE, TH, TW>;
  return ty * P::PITCH + (STRIDE == 2 ? ((tx & 1) ? P::EVENW + (tx >> 1) : (tx >> 1)) : tx);
}
// base position (tap (0, 0)) of output pixel (row, col) of the tile
template <int KS, int STRIDE, int TH, int TW>
SP_HD constexpr int out_base_pos(int row, int col) {
  using P = Patch<KS, STRIDE, TH, TW>;
  return row * STRIDE * P::PITCH + col;   // stride 2: col indexes the even plane (cc = 2 col)
}

}  // namespace sp
