// fp32 instantiations of the LDS halo-tile convolution (conv_tile_impl.h, T = F32): the reference's own arithmetic
// (AMP is commented out in the reference, Generator_3D_Resnet_ESRGAN.py:65).  This unit: outputs up to 64 channels
// (growth convs and their masked input-gradient windows, terrain convs, the z-folded last conv).
#include "conv_tile_impl.h"

int wsr_ct_run_f32_wide(CtArgs& a, int tpk, hipStream_t st);    // conv_tile_f32_wide.hip
int wsr_ct_run_f32_masked(CtArgs& a, int tpk, hipStream_t st);  // conv_tile_f32_masked.hip

template <int TPK>
static int run(CtArgs& a, hipStream_t st) {
  const int N = a.Cout;
  if (N <= 16) { pick_tile(a, 512); return launch_ct<8, 1, 4, 1, TPK, false, F32>(a, st); }
  if (N <= 32) { pick_tile(a, 512); return launch_ct<8, 1, 4, 2, TPK, false, F32>(a, st); }
  pick_tile(a, 256);
  return launch_ct<4, 1, 4, 4, TPK, false, F32>(a, st);
}

int wsr_ct_run_f32(CtArgs& a, int tpk, hipStream_t st) {
  if ((a.sx | a.sy | a.sz) != 1) return WSR_EUNSUPPORTED;
  if (a.mask_y) return wsr_ct_run_f32_masked(a, tpk, st);
  if (a.Cout > 64) return wsr_ct_run_f32_wide(a, tpk, st);
  if (tpk == 1) return run<1>(a, st);
  if (tpk == 2) return run<2>(a, st);
  return run<4>(a, st);
}
