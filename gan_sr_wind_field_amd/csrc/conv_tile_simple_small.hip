// SIMPLE instantiations (conv_tile_impl.h) of the small-volume halo-tile convs (conv_tile_small.hip: 128-voxel tiles, K-step
// shares): at the reference's own patch sizes every launch of the trunk is latency-bound, and the general kernel's prologue
// and epilogue - ~100 uniform arguments, 100-200 spilled SGPRs - are a visible part of its 13-30 us.  Same choices as
// conv_tile_small.hip; launch_ct returns WSR_EUNSUPPORTED for anything that is not a plain stride-1 conv and dispatch_ct goes
// on to the general instantiations.
#include "conv_tile_impl.h"

int wsr_ct_run_simple_small(CtArgs& a, int tpk, hipStream_t st) {
  const int N = a.Cout;
  if (tpk != 2 || WSR_ENV_INT("WSR_CT_SMALL_WK", 1) == 0) return WSR_EUNSUPPORTED;
  if (N <= 32) {
    pick_tile(a, 128);
    if (a.mask_y) return launch_ct<2, 1, 4, 2, 2, true, BF16, 4, true>(a, st);
    return launch_ct<2, 1, 4, 2, 2, false, BF16, 4, true>(a, st);
  }
  if (a.mask_y) return WSR_EUNSUPPORTED;
  if (N > 64 && N <= 256) {
    pick_tile(a, 128);
    const long tiles = (long)a.B * ((a.Xo + a.TX - 1) / a.TX) * ((a.Yo + a.TY - 1) / a.TY) * ((a.Zo + a.TZ - 1) / a.TZ);
    const int mode = WSR_ENV_INT("WSR_CT_SMALL_MODE", tiles * ((N + 127) / 128) < 160 ? 1 : 0);
    if (mode == 1) return launch_ct<2, 2, 4, 2, 2, false, BF16, 2, true>(a, st);
    return launch_ct<2, 4, 4, 2, 2, false, BF16, 1, true>(a, st);
  }
  return WSR_EUNSUPPORTED;
}
