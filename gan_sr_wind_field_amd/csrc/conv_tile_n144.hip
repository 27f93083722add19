// conv_tile_kernel<8,1,4,9,*>: the 144-channel 5x5x5 convs (hr_convs.0 forward and input gradient)
// One translation unit per output-width family: hipcc's register allocation of one template instantiation
// is perturbed by its co-compiled siblings (the 144-wide kernel lost 5 % when masked variants were added
// next to it), so the hot instantiations get a compilation unit of their own.
#define WSR_CT_XAHEAD 2  // operand requests of the K-step loop: see run_ksteps
#include "conv_tile_impl.h"

template <int TPK>
static int run(CtArgs& a, hipStream_t st) {
  const int N = a.Cout;
  (void)N;
  if (!a.mask_y && N == 144) { pick_tile(a, 512); return launch_ct<8, 1, 4, 9, TPK>(a, st); }
  return WSR_EUNSUPPORTED;
}

int wsr_ct_run_n144(CtArgs& a, int tpk, hipStream_t st) {
  if (tpk == 1) return run<1>(a, st);
  if (tpk == 2) return run<2>(a, st);
  return run<4>(a, st);
}
