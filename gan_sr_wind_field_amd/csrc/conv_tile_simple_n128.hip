// SIMPLE instantiations (conv_tile_impl.h) of the 128-wide halo-tile convs - the block-input part of a dense block's growth
// convs and the block-input window of its stacked input gradient, lr_conv - on 512- or 384-voxel tiles: the same K-step
// loop as conv_tile_n128.hip / conv_tile_tm3.hip (WSR_CT_XAHEAD 1), the general forms' run-time switches folded away.
// launch_ct returns WSR_EUNSUPPORTED for anything that is not a plain stride-1 conv and dispatch_ct goes on to the general
// instantiations.  (Its own translation unit: register allocation of the others must not move.)
#define WSR_CT_XAHEAD 1
#include "conv_tile_impl.h"

int wsr_ct_run_simple_n128(CtArgs& a, int tpk, int tm3, hipStream_t st) {
  const int N = a.Cout;
  if (tpk != 2 || a.mask_y || N <= 64 || N > 128) return WSR_EUNSUPPORTED;
  if (tm3) {
    pick_tile(a, 384);
    return launch_ct<8, 1, 3, 8, 2, false, BF16, 1, true>(a, st);
  }
  pick_tile(a, 512);
  return launch_ct<8, 1, 4, 8, 2, false, BF16, 1, true>(a, st);
}
