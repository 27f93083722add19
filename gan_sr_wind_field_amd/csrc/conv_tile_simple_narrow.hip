// SIMPLE instantiations (conv_tile_impl.h) of the 32-wide halo-tile convs: the growth convs of a dense block and their masked
// input-gradient windows, 512- or 384-voxel tiles - stride 1, no lattice, no parity phases, no split reduction, whole
// 4-channel groups.  launch_ct returns WSR_EUNSUPPORTED for anything else and dispatch_ct goes on to the general
// instantiations (conv_tile_narrow*.hip, conv_tile_tm3.hip).  (Its own translation unit: register allocation of the others
// must not move.)
#include "conv_tile_impl.h"

int wsr_ct_run_simple_narrow(CtArgs& a, int tpk, int tm3, hipStream_t st) {
  const int N = a.Cout;
  if (tpk != 2 || N <= 16 || N > 32) return WSR_EUNSUPPORTED;
  if (tm3) {
    pick_tile(a, 384);
    if (a.mask_y) return launch_ct<8, 1, 3, 2, 2, true, BF16, 1, true>(a, st);
    return launch_ct<8, 1, 3, 2, 2, false, BF16, 1, true>(a, st);
  }
  if (WSR_ENV_INT("WSR_CT_NARROW_M", 512) == 256) {  // (round 6 A/B: twice the workgroups of half the voxels - two per CU with WSR_CT_DIET=1)
    pick_tile(a, 256);
    if (a.mask_y) return launch_ct<8, 1, 2, 2, 2, true, BF16, 1, true>(a, st);
    return launch_ct<8, 1, 2, 2, 2, false, BF16, 1, true>(a, st);
  }
  pick_tile(a, 512);
  if (a.mask_y) return launch_ct<8, 1, 4, 2, 2, true, BF16, 1, true>(a, st);
  return launch_ct<8, 1, 4, 2, 2, false, BF16, 1, true>(a, st);
}
