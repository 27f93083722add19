// fp32 instantiations of the LDS halo-tile convolution with the LeakyReLU-backward mask in the epilogue (see
// conv_tile_f32.hip): the growth windows of a dense block's stacked input gradient (32 outputs), the parity input
// gradient of an up-conv (128) and the z-folded last conv's input gradient (144, with the Dropout3d keep factors).
#include "conv_tile_impl.h"

int wsr_ct_run_f32_masked(CtArgs& a, int tpk, hipStream_t st) {
  const int N = a.Cout;
  if (tpk != 2) return WSR_EUNSUPPORTED;
  if (N <= 32) { pick_tile(a, 512); return launch_ct<8, 1, 4, 2, 2, true, F32>(a, st); }
  pick_tile(a, 256);
  if (N <= 64) return launch_ct<4, 1, 4, 4, 2, true, F32>(a, st);
  if (N <= 128) return launch_ct<4, 2, 4, 4, 2, true, F32>(a, st);
  if (N <= 160) return launch_ct<4, 2, 4, 5, 2, true, F32>(a, st);
  return WSR_EUNSUPPORTED;
}
