// fp32 instantiations of the LDS halo-tile convolution, 65..256 output channels (see conv_tile_f32.hip): the
// 128-output convs of the trunk (first stage of a dense block, window 0 of its stacked input gradient, lr_conv,
// up-convs), the 5x5x5 144 -> 144 conv, and the wider windows of an un-stacked dense-block gradient.
#include "conv_tile_impl.h"

template <int TPK>
static int run(CtArgs& a, hipStream_t st) {
  const int N = a.Cout;
  if (N <= 128) { pick_tile(a, 512); return launch_ct<8, 1, 4, 8, TPK, false, F32>(a, st); }
  if (N <= 144) { pick_tile(a, 512); return launch_ct<8, 1, 4, 9, TPK, false, F32>(a, st); }
  if constexpr (TPK == 2) {
    pick_tile(a, 256);
    if (N <= 192) return launch_ct<4, 2, 4, 6, TPK, false, F32>(a, st);
    if (N <= 256) return launch_ct<4, 2, 4, 8, TPK, false, F32>(a, st);
  }
  return WSR_EUNSUPPORTED;
}

int wsr_ct_run_f32_wide(CtArgs& a, int tpk, hipStream_t st) {
  if (tpk == 1) return run<1>(a, st);
  if (tpk == 2) return run<2>(a, st);
  return run<4>(a, st);
}
