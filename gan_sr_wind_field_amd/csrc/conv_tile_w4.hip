// Halo-tile conv instantiations with FOUR waves per workgroup (one per SIMD, up to 512 registers each): a wave
// owns 8 m-tiles (128 voxels) x all n-tiles, so a K-step is 8*TN MFMAs per 8 + TN fragment reads (the 8-wave
// kernels: 4*TN per 4 + TN) - fewer LDS reads and issue slots per flop, no partner wave on the SIMD.
// Selected by WSR_CT_W4 (bit 0: 144 outputs, bit 1: 65..128, bit 2: <= 32) - a tuning switch until measured.
#include "conv_tile_impl.h"

int wsr_ct_run_w4(CtArgs& a, int tpk, int which, hipStream_t st) {
  const int N = a.Cout;
  if (a.mask_y || tpk != 2) return WSR_EUNSUPPORTED;
  if ((which & 1) && N == 144) { pick_tile(a, 512); return launch_ct<4, 1, 8, 9, 2>(a, st); }
  if ((which & 2) && N > 64 && N <= 128) { pick_tile(a, 512); return launch_ct<4, 1, 8, 8, 2>(a, st); }
  if ((which & 4) && N > 16 && N <= 32) { pick_tile(a, 512); return launch_ct<4, 1, 8, 2, 2>(a, st); }
  return WSR_EUNSUPPORTED;
}
