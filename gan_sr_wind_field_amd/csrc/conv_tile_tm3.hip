// Halo-tile conv instantiations on 384-voxel tiles (8 waves x THREE m-tiles): the trunk convs of volumes whose
// 512-voxel tiling leaves CUs idle in the one round of workgroups it launches.  The reference's cluster configuration
// (config/wind_field_GAN_3D_config_cluster.ini:42-47: batch 32 of 16 x 16 x 10 LR patches) is 6 tiles of 8 x 6 x 10 per
// sample = 192 workgroups on 256 CUs, every one of them carrying 512 MFMA rows; 8 tiles of 8 x 4 x 10 per sample are
// 256 workgroups of 384 rows - the same single round, three quarters of its length.  dispatch_ct (conv_tile.hip) sends
// a launch here only when rounds x rows-per-tile comes out smaller than on the 512-voxel tiles.
// (Its own translation unit: co-compiled instantiations perturb each other's register allocation.)
#define WSR_CT_XAHEAD 1
#include "conv_tile_impl.h"

template <int TPK>
static int run(CtArgs& a, hipStream_t st) {
  const int N = a.Cout;
  pick_tile(a, 384);
  if (N <= 32) {
    if constexpr (TPK == 2) {
      if (a.mask_y) return launch_ct<8, 1, 3, 2, TPK, true>(a, st);
    }
    if (a.mask_y) return WSR_EUNSUPPORTED;
    return launch_ct<8, 1, 3, 2, TPK>(a, st);
  }
  if (N > 64 && N <= 128 && !a.mask_y) return launch_ct<8, 1, 3, 8, TPK>(a, st);
  return WSR_EUNSUPPORTED;
}

int wsr_ct_run_tm3(CtArgs& a, int tpk, hipStream_t st) {
  if (tpk == 2) return run<2>(a, st);
  return WSR_EUNSUPPORTED;  // (the trunk's convs reduce over multiples of 16 channels)
}

// workgroups of a launch on tiles of `rows` MFMA rows (pick_tile's choice for this volume)
long wsr_ct_tiles(const CtArgs& a0, int rows) {
  CtArgs a = a0;
  pick_tile(a, rows);
  return (long)a.B * ((a.Xo + a.TX - 1) / a.TX) * ((a.Yo + a.TY - 1) / a.TY) * ((a.Zo + a.TZ - 1) / a.TZ);
}
