// 1x1x1 convolution as a streaming GEMM, second form: built for BYTES IN FLIGHT.
//
//   y[v, n] = alpha * (sum_c x[v, c] * w[n, c] + bias[n]) + beta * res[v, n] + beta2 * res2[v, n]
//
// The local-feature-fusion conv of a residual dense block (reference torch_blocks.py:278-290: 256 -> 128, its
// input gradient 128 -> 256) is ~64 flop/byte: its launch is the time to stream the activations once, and that
// is set by how many bytes the chip keeps in flight (Little: ~5 TB/s x ~2 us ~ 40 KB per CU), not by
// arithmetic.  conv_1x1.hip holds 4 waves of 512 registers per CU; this form holds 16 waves of <= 128:
//   * workgroup = 8 waves, two workgroups resident per CU (filter 64 KB of LDS each);
//   * a wave owns strips of 16 voxels; every K-step fragment of its NEXT strip (8 KB) is requested before the
//     current strip is contracted, so each wave keeps 8..16 KB in flight;
//   * the filter is staged in LDS once per workgroup with the rows of each n-tile PAIR interleaved
//     (tile 2p <- channels 32p + 8g + 0..3, tile 2p+1 <- 32p + 8g + 4..7 of lane group g), so that a lane ends
//     up with 8 consecutive output channels of its voxel: 16-byte stores, 16-byte residual / mask loads
//     (half the memory instructions of the 4-channel form);
//   * MFMA as D = W * X^T (rows = output channels), fp32 accumulators that start from bias + residuals.
#include <cstdlib>

#include "common.h"

namespace {

struct C2Args {
  const unsigned short* in;
  const unsigned short* wf;  // fragment order [K/32][n-tile][64 lanes][8] (wsr_pack_filter_frag, TPK = 1)
  unsigned short* out;
  const float* bias;
  const unsigned short* res;
  int res_ctot, res_off, res_c1;  // residual on produced channels < res_c1 only (multiple of 8)
  const unsigned short* res2;
  int res2_ctot, res2_off;
  int res2_c1;  // second residual on produced channels < res2_c1 (= res_c1 when the first one is partial, else all)
  float beta2;
  float alpha, beta, slope;
  int act;
  long nvox;
  int in_ctot, in_off, out_ctot, out_off;
  int nstrips;
  const unsigned short* mask_y;  // LeakyReLU-backward mask on produced channels [mask_c0, mask_c1), or NULL
  int mask_ctot, mask_off, mask_c0, mask_c1;
  float mask_slope;
};

__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}

// RX: the residual IS the first channels of the input window (the dense block's identity shortcut: LFF forward,
// and its in-place input gradient) - with the pair interleave lane (fr, fg) needs residual channels
// 32p + 8fg .. +7, which is exactly its own K-step fragment p: no residual loads at all.
template <int NT, int KS, bool MASK, bool RX>
__global__ __launch_bounds__(512, 4) void conv1x1_v2_kernel(const C2Args a) {
  constexpr int WAVES = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // KS*NT fragments of 1 KB, then NT*16 bias floats
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  float* btab = reinterpret_cast<float*>(smem + KS * NT * 1024);

  // this wave's strips: the first two are requested before anything else (the filter staging runs under them)
  const int stride = gridDim.x * WAVES;
  int strip = blockIdx.x * WAVES + wave;
  auto xptr = [&](int s) __attribute__((always_inline)) {
    const long v = (long)s * 16 + fr;
    return a.in + (v < a.nvox ? v : 0) * a.in_ctot + a.in_off + fg * 8;
  };
  // The filter's loads go out FIRST: loads return in order per wave, so staged after the strips' loads the filter
  // (and with it the barrier, and every store of the launch) waited until both strips of the wave had arrived - all
  // reads, then all writes (27 us for 112 MB).  Now the staging completes under the strips' flight and the stores of
  // the early strips overlap the reads of the late ones.
  constexpr int WT = KS * NT * 64 / (WAVES * 64);  // filter fragments (16 B) per thread
  static_assert(KS * NT * 64 % (WAVES * 64) == 0, "filter fragments divide over the threads");
  uint4 wtmp[WT];
  {  // n-tile pair interleave: row i of tile t' <- channel 32(t'>>1) + 8(i>>2) + 4(t'&1) + (i&3)
    const uint4* src = reinterpret_cast<const uint4*>(a.wf);
#pragma unroll
    for (int q = 0; q < WT; ++q) {
      const int idx = t + q * WAVES * 64;
      const int l = idx & 63, tp = (idx >> 6) % NT, ks = idx / (64 * NT);
      const int i = l & 15, g = l >> 4;
      const int n = 32 * (tp >> 1) + 8 * (i >> 2) + 4 * (tp & 1) + (i & 3);
      wtmp[q] = src[(ks * NT + (n >> 4)) * 64 + (n & 15) + 16 * g];
    }
  }
  uint4 xa[KS], xb[KS];
  if (strip < a.nstrips) {
    const unsigned short* p = xptr(strip);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xa[ks] = *reinterpret_cast<const uint4*>(p + ks * 32);
  }
  {  // (the second strip is requested after the filter has left its registers: 128 VGPRs at four waves per SIMD)
    uint4* dst = reinterpret_cast<uint4*>(smem);
#pragma unroll
    for (int q = 0; q < WT; ++q) dst[t + q * WAVES * 64] = wtmp[q];
  }
  __builtin_amdgcn_sched_barrier(0);
  if (strip + stride < a.nstrips) {
    const unsigned short* p = xptr(strip + stride);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xb[ks] = *reinterpret_cast<const uint4*>(p + ks * 32);
  }
  for (int k = t; k < NT * 16; k += WAVES * 64) btab[k] = a.bias ? a.bias[k] : 0.f;
  __syncthreads();
  const char* wl = smem + lane * 16;
  const float rs1 = a.res ? a.beta / a.alpha : 0.f, rs2 = a.res2 ? a.beta2 / a.alpha : 0.f;

  // One strip = 16 voxels x NT n-tiles, contracted in passes of NH <= 8 n-tiles (32 accumulator registers; the
  // strip's K-step fragments stay in registers over the passes), each pass: accumulator start, MFMAs, stores.
  constexpr int NH = NT > 8 ? 8 : NT;
  auto do_strip = [&](int s, const uint4 (&xf)[KS]) __attribute__((always_inline)) {
    const long v = (long)s * 16 + fr;
    const bool vok = v < a.nvox;
    const long vc = vok ? v : 0;
    unsigned short* o = a.out + vc * a.out_ctot + a.out_off + 8 * fg;
#pragma unroll
    for (int h0 = 0; h0 < NT; h0 += NH) {
      // accumulators start from bias + (beta/alpha) res + (beta2/alpha) res2; lane (fr, fg) owns channels
      // 32p + 8fg .. +7 of voxel v: acc[2p] the first four, acc[2p+1] the last four
      f32x4_t acc[NH];
#pragma unroll
      for (int pp = 0; pp < NH / 2; ++pp) {
        const int p = h0 / 2 + pp;
        const int co = 32 * p + 8 * fg;
        const float4 b0 = *reinterpret_cast<const float4*>(btab + co), b1 = *reinterpret_cast<const float4*>(btab + co + 4);
        float f[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        if (RX ? (p < KS && co < a.res_c1) : (a.res && co < a.res_c1)) {
          float r[8];
          if constexpr (RX) unpack8(xf[p < KS ? p : 0], r);
          else unpack8(*reinterpret_cast<const uint4*>(a.res + vc * a.res_ctot + a.res_off + co), r);
#pragma unroll
          for (int q = 0; q < 8; ++q) f[q] += rs1 * r[q];
        }
        if (a.res2 && co < a.res2_c1) {
          float r[8];
          unpack8(*reinterpret_cast<const uint4*>(a.res2 + vc * a.res2_ctot + a.res2_off + co), r);
#pragma unroll
          for (int q = 0; q < 8; ++q) f[q] += rs2 * r[q];
        }
        acc[2 * pp] = f32x4_t{f[0], f[1], f[2], f[3]};
        acc[2 * pp + 1] = f32x4_t{f[4], f[5], f[6], f[7]};
      }
      // (fenced per K-step: left alone, the scheduler requests all filter fragments up front - 256 registers -
      // and spills; NH fragments in flight are enough with 4 waves per SIMD)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        uint4 w[NH];
#pragma unroll
        for (int j = 0; j < NH; ++j) w[j] = *reinterpret_cast<const uint4*>(wl + (ks * NT + h0 + j) * 1024);
#pragma unroll
        for (int j = 0; j < NH; ++j) mma_chunk<BF16>(acc[j], w[j], xf[ks]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (vok) {
#pragma unroll
        for (int pp = 0; pp < NH / 2; ++pp) {
          const int p = h0 / 2 + pp;
          const int co = 32 * p + 8 * fg;
          float f[8] = {acc[2 * pp][0], acc[2 * pp][1], acc[2 * pp][2], acc[2 * pp][3],
                        acc[2 * pp + 1][0], acc[2 * pp + 1][1], acc[2 * pp + 1][2], acc[2 * pp + 1][3]};
          if (a.act) {
#pragma unroll
            for (int q = 0; q < 8; ++q) f[q] = f[q] > 0.f ? f[q] : f[q] * a.slope;
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) f[q] *= a.alpha;
          if constexpr (MASK) {  // (mask window bounds are multiples of 8: whole lanes are inside or outside)
            if (co >= a.mask_c0 && co < a.mask_c1) {
              const uint4 y = *reinterpret_cast<const uint4*>(a.mask_y + v * a.mask_ctot + a.mask_off + (co - a.mask_c0));
              // bf16 sign test on the raw bits: y > 0 <=> sign clear and not zero
              f[0] *= (short)(y.x & 0xFFFFu) > 0 ? 1.f : a.mask_slope; f[1] *= (int)y.x > 0xFFFF ? 1.f : a.mask_slope;
              f[2] *= (short)(y.y & 0xFFFFu) > 0 ? 1.f : a.mask_slope; f[3] *= (int)y.y > 0xFFFF ? 1.f : a.mask_slope;
              f[4] *= (short)(y.z & 0xFFFFu) > 0 ? 1.f : a.mask_slope; f[5] *= (int)y.z > 0xFFFF ? 1.f : a.mask_slope;
              f[6] *= (short)(y.w & 0xFFFFu) > 0 ? 1.f : a.mask_slope; f[7] *= (int)y.w > 0xFFFF ? 1.f : a.mask_slope;
            }
          }
          uint4 u;
          u.x = (unsigned)f2bf(f[0]) | ((unsigned)f2bf(f[1]) << 16);
          u.y = (unsigned)f2bf(f[2]) | ((unsigned)f2bf(f[3]) << 16);
          u.z = (unsigned)f2bf(f[4]) | ((unsigned)f2bf(f[5]) << 16);
          u.w = (unsigned)f2bf(f[6]) | ((unsigned)f2bf(f[7]) << 16);
          *reinterpret_cast<uint4*>(o + 32 * p) = u;
        }
      }
    }
  };

  // two strips per trip: the fragments of strip s + 2*stride are requested before strip s is contracted
  for (; strip < a.nstrips; strip += 2 * stride) {
    do_strip(strip, xa);
    if (strip + 2 * stride < a.nstrips) {
      const unsigned short* p = xptr(strip + 2 * stride);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xa[ks] = *reinterpret_cast<const uint4*>(p + ks * 32);
    }
    if (strip + stride < a.nstrips) {
      do_strip(strip + stride, xb);
      if (strip + 3 * stride < a.nstrips) {
        const unsigned short* p = xptr(strip + 3 * stride);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xb[ks] = *reinterpret_cast<const uint4*>(p + ks * 32);
      }
    }
  }
}

template <int NT, int KS, bool MASK, bool RX>
int launch_c2(C2Args& a, hipStream_t st) {
  a.nstrips = (int)((a.nvox + 15) / 16);
  const size_t lds = (size_t)KS * NT * 1024 + (size_t)NT * 16 * 4;
  auto kern = conv1x1_v2_kernel<NT, KS, MASK, RX>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  int grid = (a.nstrips + 7) / 8;
  const int cap = WSR_ENV_INT("WSR_C1_GRID", 512);  // two resident workgroups per CU (tuning aid)
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, a);
  WSR_LAUNCH_CHECK();
  return 0;
}

}  // namespace

static int wsr_conv1x1_v2_bf16(const unsigned short* in, int in_ctot, int in_off, int red, const unsigned short* wfrag,
                        unsigned short* out, int out_ctot, int out_off, int n_out, long nvox, const float* bias,
                        const unsigned short* res, int res_ctot, int res_off, int res_c1, float alpha, float beta, int act,
                        float slope, const wsr_lrelu_mask_t* mask, const unsigned short* res2, int res2_ctot, int res2_off,
                        float beta2, hipStream_t st) {
#ifdef WSR_TUNING
  if (WSR_ENV_SET("WSR_C1_V1")) return WSR_EUNSUPPORTED;  // tuning switch: the four-wave form
#endif
  if (!((red == 256 && n_out == 128) || (red == 128 && n_out == 256))) return WSR_EUNSUPPORTED;
  // 16-byte accesses on every operand: channel windows on 8-channel boundaries
  if (in_ctot % 8 || in_off % 8 || out_ctot % 8 || out_off % 8) return WSR_EUNSUPPORTED;
  if (res && (res_ctot % 8 || res_off % 8 || (res_c1 < n_out && res_c1 % 8))) return WSR_EUNSUPPORTED;
  if (res2 && (res2_ctot % 8 || res2_off % 8)) return WSR_EUNSUPPORTED;
  if (mask && (mask->y_ctot % 8 || mask->y_off % 8 || mask->c0 % 8 || mask->c1 % 8)) return WSR_EUNSUPPORTED;
  if ((res || res2) && (act || alpha == 0.f)) return WSR_EUNSUPPORTED;  // residuals are folded into the accumulator start
  if (bias && ((size_t)bias & 15)) return WSR_EUNSUPPORTED;
  C2Args a{};
  a.in = in; a.wf = wfrag; a.out = out; a.bias = bias; a.res = res;
  a.res_ctot = res_ctot; a.res_off = res_off; a.res_c1 = res_c1;
  a.res2 = res2; a.res2_ctot = res2_ctot; a.res2_off = res2_off; a.beta2 = beta2;
  // (a partial first residual - the accumulating input gradient of a dense block's LFF: the first nf channels only -
  // limits the second one to the same channels: it is the RRDB-level shortcut gradient joining the running one)
  a.res2_c1 = (res && res_c1 < n_out) ? res_c1 : 0x7FFFFFFF;
  a.alpha = alpha; a.beta = beta; a.slope = slope; a.act = act;
  a.nvox = nvox;
  a.in_ctot = in_ctot; a.in_off = in_off; a.out_ctot = out_ctot; a.out_off = out_off;
  if (mask) {
    a.mask_y = (const unsigned short*)mask->y;
    a.mask_ctot = mask->y_ctot; a.mask_off = mask->y_off;
    a.mask_c0 = mask->c0; a.mask_c1 = mask->c1;
    a.mask_slope = mask->slope;
  }
  // residual == leading channels of the input window?  (res_c1 channels of it; at most the window itself)
  const bool rx = res && res == in && res_ctot == in_ctot && res_off == in_off && !WSR_ENV_SET("WSR_C1_NORX") &&
                  (res_c1 >= n_out ? n_out <= red : res_c1 <= red);
  if (red == 256) {
    if (mask) return WSR_EUNSUPPORTED;
    return rx ? launch_c2<8, 8, false, true>(a, st) : launch_c2<8, 8, false, false>(a, st);
  }
  if (mask) return rx ? launch_c2<16, 4, true, true>(a, st) : launch_c2<16, 4, true, false>(a, st);
  return rx ? launch_c2<16, 4, false, true>(a, st) : launch_c2<16, 4, false, false>(a, st);
}

#ifdef WSR_TUNING
int wsr_conv1x1_v1_bf16(const unsigned short* in, int in_ctot, int in_off, int red, const unsigned short* wfrag,
                        unsigned short* out, int out_ctot, int out_off, int n_out, long nvox, const float* bias,
                        const unsigned short* res, int res_ctot, int res_off, int res_c1, float alpha, float beta, int act,
                        float slope, const wsr_lrelu_mask_t* mask, const unsigned short* res2, int res2_ctot, int res2_off,
                        float beta2, hipStream_t st);  // conv_1x1.hip (make TUNING=1)
#endif

// Called by the tile entry points for 1x1x1 convs; WSR_EUNSUPPORTED -> the halo-tile kernel takes over.
// `red` = reduction channels, `n_out` = produced channels.
int wsr_conv1x1_bf16(const unsigned short* in, int in_ctot, int in_off, int red, const unsigned short* wfrag,
                     unsigned short* out, int out_ctot, int out_off, int n_out, long nvox, const float* bias,
                     const unsigned short* res, int res_ctot, int res_off, int res_c1, float alpha, float beta, int act,
                     float slope, const wsr_lrelu_mask_t* mask, const unsigned short* res2, int res2_ctot, int res2_off,
                     float beta2, hipStream_t st) {
  const int rc = wsr_conv1x1_v2_bf16(in, in_ctot, in_off, red, wfrag, out, out_ctot, out_off, n_out, nvox, bias, res,
                                     res_ctot, res_off, res_c1, alpha, beta, act, slope, mask, res2, res2_ctot,
                                     res2_off, beta2, st);
#ifdef WSR_TUNING
  if (rc == WSR_EUNSUPPORTED)
    return wsr_conv1x1_v1_bf16(in, in_ctot, in_off, red, wfrag, out, out_ctot, out_off, n_out, nvox, bias, res, res_ctot,
                               res_off, res_c1, alpha, beta, act, slope, mask, res2, res2_ctot, res2_off, beta2, st);
#endif
  return rc;
}
