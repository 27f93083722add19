// Halo-tile conv instantiations: N <= 64 produced channels with the LeakyReLU-backward mask in the
// epilogue - the growth-channel windows of the stacked dense-block input gradient (its own
// translation unit: co-compiled instantiations perturb each other's register allocation).
#include "conv_tile_impl.h"

template <int TPK>
static int run(CtArgs& a, hipStream_t st) {
  const int N = a.Cout;
  if (!a.mask_y || N > 64) return WSR_EUNSUPPORTED;
  if constexpr (TPK == 2) {
    if (N <= 32) { pick_tile(a, 512); return launch_ct<8, 1, 4, 2, TPK, true>(a, st); }
    pick_tile(a, 256);
    return launch_ct<4, 1, 4, 4, TPK, true>(a, st);
  }
  return WSR_EUNSUPPORTED;
}

int wsr_ct_run_narrow_masked(CtArgs& a, int tpk, hipStream_t st) {
  if (tpk == 2) return run<2>(a, st);
  return WSR_EUNSUPPORTED;
}
