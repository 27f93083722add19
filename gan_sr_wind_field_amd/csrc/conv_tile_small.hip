// Halo-tile conv instantiations for SMALL volumes (fewer than ~128 tiles of 512 voxels, e.g. the
// reference's own 32 x 32 x 10 low-resolution patches): 128-voxel tiles, two waves per workgroup, so that
// the trunk still spreads over the chip instead of running on 20 of its 256 CUs.  Same kernel, same
// epilogues; only the tile-row count changes.
#include "conv_tile_impl.h"

template <int TPK>
static int run(CtArgs& a, hipStream_t st) {
  const int N = a.Cout;
  if constexpr (TPK == 2) {
    // K-step shares (conv_tile_impl.h, WK): the 32-wide launches on four times the waves, each walking a quarter of every
    // stage's K-steps; the 64-channel groups of the 128-wide ones on twice the waves (WSR_CT_SMALL_WK=0: two / four waves
    // per workgroup walking all of them, rounds 2-4)
    const bool wk = WSR_ENV_INT("WSR_CT_SMALL_WK", 1) != 0;
    if (a.mask_y) {
      if (N <= 32) {
        pick_tile(a, 128);
        return wk ? launch_ct<2, 1, 4, 2, TPK, true, BF16, 4>(a, st) : launch_ct<2, 1, 4, 2, TPK, true>(a, st);
      }
      return WSR_EUNSUPPORTED;
    }
    if (N <= 32) {
      pick_tile(a, 128);
      return wk ? launch_ct<2, 1, 4, 2, TPK, false, BF16, 4>(a, st) : launch_ct<2, 1, 4, 2, TPK>(a, st);
    }
    // 128 voxels x 128 channels as 2 x 4 waves of 64 x 32: four times the waves of a <2,1,4,8> tile, a quarter of
    // the work each - at these sizes the chip is latency-, not throughput-bound
    // (wider outputs - the discriminator's 256-channel layers on 16x16x64 and 8x8x64 voxels - run as groups of 128
    // channels: 64..256 workgroups instead of the 16..64 of the 256-voxel x 256-channel tile)
    if (N > 64 && N <= 256) {
      pick_tile(a, 128);
      // few tiles (the reference's 32x32x10 patches: 80 tiles): groups of 64 channels on 4 waves instead of 128 on
      // 8 - twice the workgroups, half the filter bytes each streams (measured 51.9 -> 37.1 us for 128 -> 128)
      const long tiles = (long)a.B * ((a.Xo + a.TX - 1) / a.TX) * ((a.Yo + a.TY - 1) / a.TY) * ((a.Zo + a.TZ - 1) / a.TZ);
      const int mode = WSR_ENV_INT("WSR_CT_SMALL_MODE", tiles * ((N + 127) / 128) < 160 ? 1 : 0);
      // (four shares = 16 waves of <= 128 registers: the instantiation spills - two shares it is)
      if (mode == 1) return wk ? launch_ct<2, 2, 4, 2, TPK, false, BF16, 2>(a, st) : launch_ct<2, 2, 4, 2, TPK>(a, st);
      return launch_ct<2, 4, 4, 2, TPK>(a, st);
    }
  }
  return WSR_EUNSUPPORTED;
}

int wsr_ct_run_small(CtArgs& a, int tpk, hipStream_t st) {
  if (tpk == 2) return run<2>(a, st);
  return WSR_EUNSUPPORTED;
}
