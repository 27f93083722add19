// Halo-tile conv instantiations for STRIDED forward convs (the discriminator's (4,4,k) stride (2,2,1|2)
// down-sampling layers, torch_blocks.py:138-142): 256-voxel output tiles - the input halo of a stride-2
// tile is 4..8x its output volume - as 8 waves x 2 m-tiles.
#include "conv_tile_impl.h"

template <int TPK>
static int run(CtArgs& a, hipStream_t st) {
  const int N = a.Cout;
  if (a.mask_y || a.ups) return WSR_EUNSUPPORTED;
  if constexpr (TPK == 2) {
    // the halo of a strided tile grows with the stride and the filter extent: fall back to flatter tiles
    // until the image fits the per-wave DMA budget and LDS
    static const int shapes[][3] = {{4, 4, 16}, {4, 8, 8}, {4, 4, 8}, {2, 4, 8}, {2, 2, 8}};
    for (const auto& sh : shapes) {
      a.TX = sh[0]; a.TY = sh[1]; a.TZ = sh[2] < a.Zo ? sh[2] : a.Zo;
      int rc;
      // few output tiles (the deep layers: 16x16x64 .. 4x4x32 voxels): narrower channel groups, more workgroups
      const long tiles = (long)a.B * ((a.Xo + a.TX - 1) / a.TX) * ((a.Yo + a.TY - 1) / a.TY) * ((a.Zo + a.TZ - 1) / a.TZ);
      const bool few = !WSR_ENV_SET("WSR_CT_STRIDED_WIDE") && tiles * ((N + 127) / 128) < 128;
      if (N <= 32 || (few && tiles * ((N + 63) / 64) < 128)) rc = launch_ct<8, 1, 2, 2, TPK>(a, st);
      else if (N <= 64 || few) rc = launch_ct<8, 1, 2, 4, TPK>(a, st);
      else rc = launch_ct<8, 1, 2, 8, TPK>(a, st);  // wider outputs: groups of 128 channels
      if (rc != WSR_EUNSUPPORTED) return rc;
    }
  }
  return WSR_EUNSUPPORTED;
}

int wsr_ct_run_strided(CtArgs& a, int tpk, hipStream_t st) {
  if (tpk == 2) return run<2>(a, st);
  return WSR_EUNSUPPORTED;
}
