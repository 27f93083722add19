// 1x1x1 convolution (the local-feature-fusion conv of a residual dense block, reference
// torch_blocks.py:278-290, and its input gradient) as a streaming GEMM, bf16:
//
//   y[v, n] = alpha * (sum_c x[v, c] * w[n, c] + bias[n]) + beta * res[v, n]
//
// At 256 -> 128 channels the arithmetic intensity is ~64 flop/byte: the launch is bound by streaming
// the activations once from HBM, not by the MFMAs.  So the structure is the opposite of the halo-tile
// kernel: the whole filter (<= 64 KB in MFMA-fragment order) is staged in LDS ONCE per workgroup and
// re-used for every voxel the workgroup visits, while the activation fragments - each is needed by
// exactly one wave - go from global memory straight to registers (16 voxels x 64 contiguous bytes per
// wave-instruction), all K-steps of a 32-voxel strip in flight at once.  MFMA is issued as
// D = W * X^T, so a lane holds 4 consecutive output channels of one voxel (8-byte stores).
#include <cstdlib>

#include "common.h"

namespace {

struct C1Args {
  const unsigned short* in;
  const unsigned short* wf;  // fragment order [K/32][n-tile][64 lanes][8] (wsr_pack_filter_frag, TPK = 1)
  unsigned short* out;
  const float* bias;
  const unsigned short* res;
  int res_ctot, res_off, res_c1;  // residual on produced channels < res_c1 only
  const unsigned short* res2;     // second residual (all produced channels), or NULL
  int res2_ctot, res2_off;
  float beta2;
  float alpha, beta, slope;
  int act;
  long nvox;
  int in_ctot, in_off, out_ctot, out_off, Cout;
  int nstrips;  // ceil(nvox / (16*TM))
  const unsigned short* mask_y;  // LeakyReLU-backward mask on produced channels [mask_c0, mask_c1), or NULL
  int mask_ctot, mask_off, mask_c0, mask_c1;
  float mask_slope;
};

// NH = 2: the produced channels are split in two halves of TN n-tiles handled by different waves of the
// workgroup (both read the same activation strip): half the accumulators per wave, room for the prefetch.
template <int TN, int TM, int KS, bool MASK, int WAVES, bool PF, int NH>
__global__ __launch_bounds__(WAVES * 64) void conv1x1_kernel(const C1Args a) {
  constexpr int TNT = TN * NH, SW = WAVES / NH;  // n-tiles of the filter, waves per column half
  extern __shared__ __attribute__((aligned(16))) char smem[];  // KS * TNT fragments of 1 KB
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  {  // stage the filter: linear copy of KS*TN KB
    const uint4* src = reinterpret_cast<const uint4*>(a.wf);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    for (int i = t; i < KS * TNT * 64; i += WAVES * 64) dst[i] = src[i];
  }
  __syncthreads();
  const char* wl = smem + lane * 16;
  const int jb = (wave / SW) * TN;  // first n-tile of this wave's column half
  const int swave = wave % SW;

  // Per strip: every load (activation fragments, bias, residual, mask values) is requested before the first
  // store - a load placed after a store cannot be hoisted by the compiler (possible aliasing) and would
  // cost a memory round trip per tile.  (Prefetching the next strip's fragments into a second register set
  // was measured slower: 128 more VGPRs spill.)
  const int stride = gridDim.x * SW;
  auto load_x = [&](int strip, uint4 (&xf)[KS][TM]) {
    const long v0 = (long)strip * (16 * TM);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long v = v0 + 16 * i + fr;
      const unsigned short* p = a.in + (v < a.nvox ? v : 0) * a.in_ctot + a.in_off + fg * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xf[ks][i] = *reinterpret_cast<const uint4*>(p + ks * 32);
    }
  };
  auto do_strip = [&](int strip, const uint4 (&xf)[KS][TM]) {
    const long v0 = (long)strip * (16 * TM);
    // The accumulators start from bias + (beta/alpha) * residual (this kernel takes a residual only without
    // an activation), so neither needs registers of its own next to the two fragment sets; the mask values
    // of the strip (input-gradient launches) are requested here, before any store.
    f32x4_t acc[TM][TN];
    uint2 yy[TM][MASK ? TN : 1];
    const float rscale = a.res ? a.beta / a.alpha : 0.f, rscale2 = a.res2 ? a.beta2 / a.alpha : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long v = v0 + 16 * i + fr;
      const bool vok = v < a.nvox;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int co0 = 16 * (jb + j) + 4 * fg;
        const bool ok = vok && co0 < a.Cout;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias && co0 < a.Cout) b4 = *reinterpret_cast<const float4*>(a.bias + co0);
        if (a.res && ok && co0 < a.res_c1) {
          const uint2 r2 = *reinterpret_cast<const uint2*>(a.res + v * a.res_ctot + a.res_off + co0);
          b4.x += rscale * bf2f((unsigned short)(r2.x & 0xFFFFu));
          b4.y += rscale * bf2f((unsigned short)(r2.x >> 16));
          b4.z += rscale * bf2f((unsigned short)(r2.y & 0xFFFFu));
          b4.w += rscale * bf2f((unsigned short)(r2.y >> 16));
        }
        if (a.res2 && ok) {
          const uint2 r2 = *reinterpret_cast<const uint2*>(a.res2 + v * a.res2_ctot + a.res2_off + co0);
          b4.x += rscale2 * bf2f((unsigned short)(r2.x & 0xFFFFu));
          b4.y += rscale2 * bf2f((unsigned short)(r2.x >> 16));
          b4.z += rscale2 * bf2f((unsigned short)(r2.y & 0xFFFFu));
          b4.w += rscale2 * bf2f((unsigned short)(r2.y >> 16));
        }
        acc[i][j] = f32x4_t{b4.x, b4.y, b4.z, b4.w};
        if constexpr (MASK) {
          yy[i][j] = make_uint2(0x3F803F80u, 0x3F803F80u);  // +1: derivative 1 outside the mask window
          if (ok && co0 >= a.mask_c0 && co0 < a.mask_c1)
            yy[i][j] = *reinterpret_cast<const uint2*>(a.mask_y + v * a.mask_ctot + a.mask_off + (co0 - a.mask_c0));
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const uint4 w = *reinterpret_cast<const uint4*>(wl + (ks * TNT + jb + j) * 1024);
#pragma unroll
        for (int i = 0; i < TM; ++i) mma_chunk<BF16>(acc[i][j], w, xf[ks][i]);
      }
    }
    // ---- epilogue: acc[i][j][r] -> channel 16j + 4fg + r of voxel v0 + 16i + fr
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long v = v0 + 16 * i + fr;
      if (v >= a.nvox) continue;
      unsigned short* o = a.out + v * a.out_ctot + a.out_off + 4 * fg;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int co0 = 16 * (jb + j) + 4 * fg;
        if (co0 >= a.Cout) continue;  // Cout is a multiple of 4 here (checked on the host)
        float4 o4 = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if (a.act) {
          o4.x = o4.x > 0.f ? o4.x : o4.x * a.slope;
          o4.y = o4.y > 0.f ? o4.y : o4.y * a.slope;
          o4.z = o4.z > 0.f ? o4.z : o4.z * a.slope;
          o4.w = o4.w > 0.f ? o4.w : o4.w * a.slope;
        }
        o4.x *= a.alpha; o4.y *= a.alpha; o4.z *= a.alpha; o4.w *= a.alpha;
        if constexpr (MASK) {  // bf16 sign test on the raw bits: y > 0 <=> sign clear and not zero
          o4.x *= (short)(yy[i][j].x & 0xFFFFu) > 0 ? 1.f : a.mask_slope;
          o4.y *= (int)yy[i][j].x > 0xFFFF ? 1.f : a.mask_slope;
          o4.z *= (short)(yy[i][j].y & 0xFFFFu) > 0 ? 1.f : a.mask_slope;
          o4.w *= (int)yy[i][j].y > 0xFFFF ? 1.f : a.mask_slope;
        }
        st4<BF16>(o + 16 * (jb + j), o4);
      }
    }
  };
  if constexpr (PF) {  // the next strip's activation fragments are requested before this strip is computed
    uint4 xa[KS][TM], xb[KS][TM];
    int strip = blockIdx.x * SW + swave;
    if (strip < a.nstrips) load_x(strip, xa);
    for (; strip < a.nstrips; strip += 2 * stride) {
      if (strip + stride < a.nstrips) load_x(strip + stride, xb);
      do_strip(strip, xa);
      if (strip + stride < a.nstrips) {
        if (strip + 2 * stride < a.nstrips) load_x(strip + 2 * stride, xa);
        do_strip(strip + stride, xb);
      }
    }
  } else {
    for (int strip = blockIdx.x * SW + swave; strip < a.nstrips; strip += stride) {
      uint4 xa[KS][TM];
      load_x(strip, xa);
      do_strip(strip, xa);
    }
  }
}

template <int TN, int TM, int KS, bool MASK, bool PF = false, int NH = 1>
int launch_c1(const C1Args& a0, hipStream_t st) {
  C1Args a = a0;
  a.nstrips = (int)((a.nvox + 16 * TM - 1) / (16 * TM));
  const size_t lds = (size_t)KS * TN * NH * 1024;
  constexpr int WAVES = 4;  // (8 waves halve the register budget: the up-front loads of a strip then spill)
  auto kern = conv1x1_kernel<TN, TM, KS, MASK, WAVES, PF, NH>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  int grid = (a.nstrips + WAVES / NH - 1) / (WAVES / NH);
  const int cap = WSR_ENV_INT("WSR_C1_GRID", 256);  // one workgroup per CU measured best (tuning aid)
  if (grid > cap) grid = cap;  // the filter is staged once per workgroup
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVES * 64), lds, st, a);
  WSR_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// The superseded four-wave form (round 1).  Built only by `make TUNING=1`; conv_1x1_v2.hip's dispatcher falls back
// to it for the shapes the 16-wave form does not cover and under WSR_C1_V1.
int wsr_conv1x1_v1_bf16(const unsigned short* in, int in_ctot, int in_off, int red, const unsigned short* wfrag,
                        unsigned short* out, int out_ctot, int out_off, int n_out, long nvox, const float* bias,
                        const unsigned short* res, int res_ctot, int res_off, int res_c1, float alpha, float beta, int act,
                        float slope, const wsr_lrelu_mask_t* mask, const unsigned short* res2, int res2_ctot, int res2_off,
                        float beta2, hipStream_t st) {
  if (red % 32 || red > 256 || n_out % 4 || n_out > 256) return WSR_EUNSUPPORTED;
  if (in_ctot % 8 || in_off % 8 || out_ctot % 4 || out_off % 4 || (res && (res_ctot % 4 || res_off % 4)))
    return WSR_EUNSUPPORTED;
  C1Args a{};
  a.in = in; a.wf = wfrag; a.out = out; a.bias = bias; a.res = res;
  a.res_ctot = res_ctot; a.res_off = res_off; a.res_c1 = res_c1;
  a.res2 = res2; a.res2_ctot = res2_ctot; a.res2_off = res2_off; a.beta2 = beta2;
  if (res2 && (res2_ctot % 4 || res2_off % 4)) return WSR_EUNSUPPORTED;
  if (res2 && res && res_c1 < n_out) return WSR_EUNSUPPORTED;  // (partial second residual: the 16-wave form only)
  a.alpha = alpha; a.beta = beta; a.slope = slope; a.act = act;
  a.nvox = nvox;
  a.in_ctot = in_ctot; a.in_off = in_off; a.out_ctot = out_ctot; a.out_off = out_off; a.Cout = n_out;
  if (mask) {
    a.mask_y = (const unsigned short*)mask->y;
    a.mask_ctot = mask->y_ctot; a.mask_off = mask->y_off;
    a.mask_c0 = mask->c0; a.mask_c1 = mask->c1;
    a.mask_slope = mask->slope;
  }
  if (n_out % 16) return WSR_EUNSUPPORTED;  // the fragment-order filter must have exactly TN n-tiles
  const int nt = n_out / 16, ks = red / 32;
  if ((a.res || a.res2) && (a.act || a.alpha == 0.f)) return WSR_EUNSUPPORTED;  // residuals are folded into the accumulator start
  if (a.bias && ((size_t)a.bias & 15)) return WSR_EUNSUPPORTED;
  if (!mask) {
    if (nt == 8 && ks == 8) {  // 256 -> 128 (LFF forward): 16-voxel strips, next strip prefetched (-18 % vs <8,2,8>)
      if (WSR_ENV_SET("WSR_C1_NOPF")) return launch_c1<8, 2, 8, false>(a, st);
      return launch_c1<8, 1, 8, false, true>(a, st);
    }
    if (nt == 16 && ks == 4) return launch_c1<16, 1, 4, false>(a, st);  // 128 -> 256
    if (nt == 8 && ks == 4) return launch_c1<8, 2, 4, false>(a, st);    // 128 -> 128
  } else {
    // 128 -> 256 (LFF input gradient).  (Two column halves on different waves + prefetch, launch_c1<8,1,4,true,true,2>,
    // was measured 11 % slower; prefetch alone spills.)
    if (nt == 16 && ks == 4) return launch_c1<16, 1, 4, true>(a, st);
    if (nt == 8 && ks == 4) return launch_c1<8, 2, 4, true>(a, st);
  }
  return WSR_EUNSUPPORTED;
}
