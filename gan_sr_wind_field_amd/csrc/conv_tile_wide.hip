// conv_tile_kernel for 65..256 output channels (128-channel convs, dense-block input gradients)
// One translation unit per output-width family: hipcc's register allocation of one template instantiation
// is perturbed by its co-compiled siblings (the 144-wide kernel lost 5 % when masked variants were added
// next to it), so the hot instantiations get a compilation unit of their own.
#include "conv_tile_impl.h"

template <int TPK>
static int run(CtArgs& a, hipStream_t st) {
  const int N = a.Cout;
  (void)N;
  if (a.mask_y) return WSR_EUNSUPPORTED;
  if (N <= 128) { pick_tile(a, 256); return launch_ct<4, 2, 4, 4, TPK>(a, st); }
  if (N <= 160) { pick_tile(a, 256); return launch_ct<4, 2, 4, 5, TPK>(a, st); }
  if (N <= 192) { pick_tile(a, 256); return launch_ct<4, 2, 4, 6, TPK>(a, st); }
  if (N <= 224) { pick_tile(a, 256); return launch_ct<4, 2, 4, 7, TPK>(a, st); }
  if (N <= 256) { pick_tile(a, 256); return launch_ct<4, 2, 4, 8, TPK>(a, st); }
  return WSR_EUNSUPPORTED;
}

int wsr_ct_run_wide(CtArgs& a, int tpk, hipStream_t st) {
  if (tpk == 1) return run<1>(a, st);
  if (tpk == 2) return run<2>(a, st);
  return run<4>(a, st);
}
