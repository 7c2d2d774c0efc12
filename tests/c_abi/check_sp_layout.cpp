// Host-side check of disconet_amd/csrc/sp_layout.h (no GPU): for every tile shape of
// conv_sp.hip's menu
//   * the LDS image map patch_pos / patch_col_of is a bijection between patch pixels and
//     non-padding positions;
//   * an output pixel's tap read (out_base_pos + tap_offset) lands on the position of the
//     input pixel the convolution needs;
//   * the lane -> pixel map covers the tile exactly once;
//   * every 16-lane ds_read_b128 service group touches 16 distinct 16-byte slots (mod 16).
// Build: g++ -std=c++17 -I disconet_amd/csrc tests/c_abi/check_sp_layout.cpp -o check_sp_layout
#include <cstdio>
#include <set>
#include <vector>
#include "sp_layout.h"

static int g_bad = 0;
#define EXPECT(c, ...) do { if (!(c)) { ++g_bad; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

template <int KS, int STRIDE, int TH, int TW>
void check(const char* name, int groups) {
  using P = sp::Patch<KS, STRIDE, TH, TW>;
  // bijection
  std::set<int> seen;
  for (int r = 0; r < P::PH; ++r)
    for (int cc = 0; cc < P::PW; ++cc) {
      const int pos = sp::patch_pos<KS, STRIDE, TH, TW>(r, cc);
      EXPECT(pos >= 0 && pos < P::NPIX, "%s: pos %d out of plane", name, pos);
      EXPECT(seen.insert(pos).second, "%s: position %d used twice", name, pos);
      EXPECT(pos / P::PITCH == r, "%s: row of pos", name);
      EXPECT((sp::patch_col_of<KS, STRIDE, TH, TW>(pos % P::PITCH)) == cc, "%s: inverse col (%d,%d)", name, r, cc);
    }
  for (int pos = 0; pos < P::PITCH; ++pos) {
    const int cc = sp::patch_col_of<KS, STRIDE, TH, TW>(pos);
    EXPECT(cc < P::PW, "%s: inverse beyond patch", name);
    if (cc >= 0) EXPECT((sp::patch_pos<KS, STRIDE, TH, TW>(0, cc)) == pos, "%s: inverse mismatch at %d", name, pos);
  }
  // lane -> pixel coverage and tap reads
  std::set<int> pix;
  for (int gm = 0; gm < groups; ++gm)
    for (int j = 0; j < 32; ++j) {
      const int row = sp::tile_row<TW>(gm, j), col = sp::tile_col<TW>(j);
      EXPECT(row >= 0 && row < TH && col >= 0 && col < TW, "%s: lane pixel (%d,%d) outside tile", name, row, col);
      EXPECT(pix.insert(row * TW + col).second, "%s: pixel (%d,%d) owned twice", name, row, col);
      for (int ty = 0; ty < KS; ++ty)
        for (int tx = 0; tx < KS; ++tx) {
          const int want = sp::patch_pos<KS, STRIDE, TH, TW>(row * STRIDE + ty, col * STRIDE + tx);
          const int got = sp::out_base_pos<KS, STRIDE, TH, TW>(row, col) + sp::tap_offset<KS, STRIDE, TH, TW>(ty, tx);
          EXPECT(want == got, "%s: tap (%d,%d) of pixel (%d,%d): %d != %d", name, ty, tx, row, col, got, want);
        }
    }
  EXPECT((int)pix.size() == TH * TW, "%s: %zu of %d pixels covered", name, pix.size(), TH * TW);
  // bank conflicts: service groups of ds_read_b128 within 32 lanes
  for (int gm = 0; gm < groups; ++gm)
    for (int ty = 0; ty < KS; ++ty)
      for (int tx = 0; tx < KS; ++tx)
        for (int g2 = 0; g2 < 2; ++g2) {
          std::set<int> slots;
          for (int j = 0; j < 32; ++j)
            if (sp::in_g2(j) == (g2 == 1))
              slots.insert((sp::out_base_pos<KS, STRIDE, TH, TW>(sp::tile_row<TW>(gm, j), sp::tile_col<TW>(j)) +
                            sp::tap_offset<KS, STRIDE, TH, TW>(ty, tx)) % 16);
          EXPECT(slots.size() == 16, "%s: group %d tap (%d,%d) half %d: %zu distinct slots", name, gm, ty, tx, g2,
                 slots.size());
        }
  printf("%-22s PH %2d PW %2d pitch %2d plane %4d pieces\n", name, P::PH, P::PW, P::PITCH, P::NPIX);
}

int main() {
  int ranks = 0;
  for (int j = 0; j < 32; ++j) ranks += sp::rank16(j);
  EXPECT(ranks == 2 * 120, "rank16 is not a permutation of 0..15 per group");
  check<3, 1, 8, 32>("3x3 s1 8x32", 8);
  check<3, 1, 8, 16>("3x3 s1 8x16", 4);
  check<3, 1, 8, 8>("3x3 s1 8x8", 2);
  check<3, 2, 8, 16>("3x3 s2 8x16", 4);
  check<3, 2, 8, 8>("3x3 s2 8x8", 2);
  check<1, 1, 8, 32>("1x1 8x32", 8);
  check<1, 1, 8, 8>("1x1 8x8", 2);
  check<3, 1, 16, 32>("3x3 s1 16x32", 16);
  printf(g_bad ? "SP LAYOUT CHECK FAILED (%d)\n" : "SP LAYOUT CHECK PASSED\n", g_bad);
  return g_bad ? 1 : 0;
}
