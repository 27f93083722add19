// Halo-tile conv instantiation: 65..128 produced channels on a 512-voxel tile (8 waves x 4 m-tiles x 8
// n-tiles).  At the trunk's resolution (32 x 32 x 128 voxels) that is one workgroup per CU in a single
// round; the 256-voxel tiles of conv_tile_wide.hip need two rounds there, pay the prologue and epilogue
// twice and stream every filter stage twice as often per flop.
#define WSR_CT_XAHEAD 1  // operand requests of the K-step loop: see run_ksteps
#include "conv_tile_impl.h"

template <int TPK>
static int run(CtArgs& a, hipStream_t st) {
  const int N = a.Cout;
  if (a.mask_y || N <= 64 || N > 128) return WSR_EUNSUPPORTED;
  pick_tile(a, 512);
  return launch_ct<8, 1, 4, 8, TPK>(a, st);
}

int wsr_ct_run_n128(CtArgs& a, int tpk, hipStream_t st) {
  if (tpk == 1) return run<1>(a, st);
  if (tpk == 2) return run<2>(a, st);
  return run<4>(a, st);
}
