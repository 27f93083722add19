// 32-wide halo-tile convs (the growth convs of a dense block and their masked input-gradient windows) on SIXTEEN waves:
// two K-step shares per (m-tiles, n-tiles) - conv_tile_impl.h, WK.  At the trunk's resolution these launches are one round
// of 256 workgroups whose matrix pipe is busy 0.15-0.18 of the time: each wave's K-step is 8 MFMAs behind ~25 instructions
// of address arithmetic and six LDS fragment reads, and two waves per SIMD do not cover each other's waits.  With the
// kernel's 125 registers a CU holds four waves per SIMD: the same tile, the K-steps of every weight stage dealt to two
// shares, partial sums joined in LDS (fixed order).  (Its own translation unit: register allocation of the 8-wave
// instantiations must not move.)
#include "conv_tile_impl.h"

template <int TPK>
static int run(CtArgs& a, hipStream_t st) {
  const int N = a.Cout;
  if (N <= 16 || N > 32) return WSR_EUNSUPPORTED;
  pick_tile(a, 512);
  if constexpr (TPK == 2) {
    if (a.mask_y) return launch_ct<8, 1, 4, 2, TPK, true, BF16, 2>(a, st);
  }
  if (a.mask_y) return WSR_EUNSUPPORTED;
  return launch_ct<8, 1, 4, 2, TPK, false, BF16, 2>(a, st);
}

int wsr_ct_run_narrow_wk(CtArgs& a, int tpk, hipStream_t st) {
  if (tpk == 2) return run<2>(a, st);
  return WSR_EUNSUPPORTED;
}
