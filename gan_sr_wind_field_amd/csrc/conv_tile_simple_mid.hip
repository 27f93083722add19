// SIMPLE instantiations (conv_tile_impl.h) of the 64- and 96-wide halo-tile convs on 512-voxel tiles: the stages of a dense
// block's forward grouped by SOURCE window (engine.conv_dense, round 6) - the growth channels conv j - 1 just produced are
// contracted into the windows of ALL later convs of the block at once (96, 64, 32 outputs at a reduction of 32 channels x
// 27 taps) instead of every conv re-reading every earlier window (32 outputs at 32, 64, 96 channels).  Same arithmetic;
// a weight-stage phase of the 32-wide launches holds 1.5 us of MFMAs against ~3 us for its DMA round trip (their main loop
// is latency-bound: 6.4 / 12.8 / 19.2 us for 2 / 4 / 6 phases), a 96-wide phase 4 us.  Only launches that carry the
// grouped stages' epilogue (act = 2 with a partial activation window) come here; everything else keeps its kernel.
// (Its own translation unit: register allocation of the others must not move.)
#include "conv_tile_impl.h"

int wsr_ct_run_simple_mid(CtArgs& a, int tpk, hipStream_t st) {
  const int N = a.Cout;
  if (tpk != 2 || a.mask_y || N <= 32 || N > 96) return WSR_EUNSUPPORTED;
  pick_tile(a, 512);
  if (N <= 64) return launch_ct<8, 1, 4, 4, 2, false, BF16, 1, true>(a, st);
  return launch_ct<8, 1, 4, 6, 2, false, BF16, 1, true>(a, st);
}
