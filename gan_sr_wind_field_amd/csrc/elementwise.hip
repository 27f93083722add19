// HBM-bound helper kernels around the convolutions: filter packing, LeakyReLU
// backward, channel-window axpby, nearest-upsample backward, planar<->NDHWC,
// BatchNorm3d statistics/apply/backward and the flat Adam step.  All are
// grid-stride, vectorised where the channel window allows it.
#include "common.h"
#include "stencil.h"

std::atomic<int> g_wsr_env_gen{0};  // generation of the cached environment switches (common.h)

extern "C" int wsr_reload_env(void) {
  g_wsr_env_gen.fetch_add(1, std::memory_order_acq_rel);
  return 0;
}

namespace {

constexpr int EW_BLOCK = 256;
static inline int ew_grid(long n) {
  long g = (n + EW_BLOCK - 1) / EW_BLOCK;
  if (g > 256 * 8) g = 256 * 8;  // 8 workgroups per CU, grid-stride the rest
  if (g < 1) g = 1;
  return (int)g;
}

// ---- filter packing -------------------------------------------------------------
template <class T>
__global__ void pack_filter_kernel(const float* __restrict__ w, typename T::elem* __restrict__ out, int Cout, int taps,
                                   int Cin, int transpose, int kpad) {
  const int rows = transpose ? Cin : Cout;
  const long total = (long)rows * taps * kpad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % kpad);
    const long rt = i / kpad;
    const int tap = (int)(rt % taps);
    const int r = (int)(rt / taps);
    float v = 0.f;  // master is the logical nn.Conv3d layout [Cout][Cin][taps]
    if (!transpose) {
      if (k < Cin) v = w[((long)r * Cin + k) * taps + tap];
    } else {
      if (k < Cout) v = w[((long)k * Cin + r) * taps + tap];
    }
    stf<T>(out + i, v);
  }
}

// dst [Cout][Cin][taps] (logical nn.Conv3d layout) += src [Cout][taps][kpad]
__global__ void unpack_wgrad_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int taps, int Cin,
                                    int kpad, float scale, int accumulate) {
  const long total = (long)Cout * Cin * taps;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const long nc = i / taps;
    const int c = (int)(nc % Cin);
    const int n = (int)(nc / Cin);
    const float v = scale * src[((long)n * taps + tap) * kpad + c];
    dst[i] = accumulate ? dst[i] + v : v;
  }
}

// many filter gradients in one launch: one block row (blockIdx.y) per job
__global__ void unpack_wgrad_multi_kernel(const wsr_unpack_job_t* __restrict__ jobs) {
  const wsr_unpack_job_t j = jobs[blockIdx.y];
  const long total = (long)j.Cout * j.Cin * j.taps;
  const float* __restrict__ src = j.src;
  float* __restrict__ dst = j.dst;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % j.taps);
    const long nc = i / j.taps;
    const int c = (int)(nc % j.Cin);
    const int n = (int)(nc / j.Cin);
    const float v = j.scale * src[((long)n * j.taps + tap) * j.kpad + c];
    dst[i] = j.accumulate ? dst[i] + v : v;
  }
}

// Deterministic filter gradients: the sum over the n_parts split copies (fixed order) and the move to the master
// layout in one pass.  One workgroup per (job, output channel n, 64 input channels): the packed rows
// [tap][c0 .. c0+63] of every part are read coalesced, summed, transposed through LDS and written as the
// contiguous run dst[n][c0 .. c0+63][taps].
// sixteen / eight / four copies in flight per thread (the copies are megabytes apart: every load is an HBM round
// trip), added in index order
__device__ __forceinline__ float4 ur_sum_parts(const float* p, int nparts, long part_stride) {
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = 0;
  for (; s + 16 <= nparts; s += 16) {
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4*>(p + (long)(s + u) * part_stride);
#pragma unroll
    for (int u = 0; u < 16; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
  }
  if (s + 8 <= nparts) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (long)(s + u) * part_stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    s += 8;
  }
  if (s + 4 <= nparts) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(p + (long)(s + u) * part_stride);
#pragma unroll
    for (int u = 0; u < 4; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    s += 4;
  }
  for (; s < nparts; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(p + (long)s * part_stride);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  return a;
}

constexpr int UR_SH = 12288;  // floats of LDS per workgroup (48 KB)

__global__ __launch_bounds__(512) void unpack_reduce_multi_kernel(const wsr_unpack_job_t* __restrict__ jobs, int rows_ok) {
  __shared__ float shf[UR_SH];
  float (*sh)[129] = reinterpret_cast<float (*)[129]>(shf);  // chunk form: [c 64][tap], taps <= 128
  const wsr_unpack_job_t j = jobs[blockIdx.y];
  const int cchunks = (j.Cin + 63) / 64;
  // Row form (round 4): one workgroup per output channel n takes the WHOLE packed row [tap][0 .. Cin) of every copy -
  // runs of Cin * 4 bytes (the rows of consecutive taps touch when kpad = Cin) instead of the chunk form's 256-byte
  // pieces, which held the pass at 2.5 TB/s - and turns it through LDS as [c][tap | 1].  Same additions in the same
  // order: bit-identical results.  Jobs whose [Cin][taps] image does not fit 48 KB (the 5x5x5 conv) keep the chunk form.
  {
    const int tp = j.taps | 1;
    const bool vec = (j.kpad & 3) == 0 && (j.part_stride & 3) == 0 && (((size_t)j.src) & 15) == 0;
    if (rows_ok && vec && j.n_parts > 0 && (long)j.Cin * tp <= UR_SH) {
      const int cq = (j.Cin + 3) >> 2;        // float4 pieces per tap (inside the row: kpad is a multiple of 4)
      const int nq = j.taps * cq;
      for (int n = blockIdx.x; n < j.Cout; n += gridDim.x) {
        __syncthreads();
        const float* base = j.src + (long)n * j.taps * j.kpad;
        for (int q = threadIdx.x; q < nq; q += 512) {
          const int tap = q / cq, c = (q - tap * cq) * 4;
          const float4 a = ur_sum_parts(base + (long)tap * j.kpad + c, j.n_parts, j.part_stride);
          shf[c * tp + tap] = a.x;
          if (c + 1 < j.Cin) shf[(c + 1) * tp + tap] = a.y;
          if (c + 2 < j.Cin) shf[(c + 2) * tp + tap] = a.z;
          if (c + 3 < j.Cin) shf[(c + 3) * tp + tap] = a.w;
        }
        __syncthreads();
        float* d = j.dst + (long)n * j.Cin * j.taps;
        for (int idx = threadIdx.x; idx < j.Cin * j.taps; idx += 512) {
          const int c = idx / j.taps, tap = idx - c * j.taps;
          const float v = j.scale * shf[c * tp + tap];
          d[idx] = j.accumulate ? d[idx] + v : v;
        }
      }
      return;
    }
  }
  // 16 lanes x float4 = 64 channels, 32 taps at a time (a 3x3x3 filter: one pass); a thread's chain is
  // ceil(n_parts / 16) dependent round trips - at 8 in flight and 16 taps per pass the 64-copy jobs were latency-bound
  const int c4 = (threadIdx.x & 15) * 4, tl = threadIdx.x >> 4;
  const int nparts = j.n_parts > 0 ? j.n_parts : 1;
  const bool vec = (j.kpad & 3) == 0 && (j.part_stride & 3) == 0 && (((size_t)j.src) & 15) == 0;
  for (int item = blockIdx.x; item < j.Cout * cchunks; item += gridDim.x) {
    const int n = item / cchunks, c0 = (item - n * cchunks) * 64;
    const int cw = min(64, j.Cin - c0);
    __syncthreads();
    for (int tap = tl; tap < j.taps; tap += 32) {
      const float* p = j.src + ((long)n * j.taps + tap) * j.kpad + c0 + c4;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      if (vec && c0 + c4 + 4 <= j.kpad) {  // (the row is kpad long: reading past Cin inside it is harmless)
        a = ur_sum_parts(p, nparts, j.part_stride);
      } else {
        for (int s = 0; s < nparts; ++s) {
          const float* q = p + (long)s * j.part_stride;
          if (c4 + 0 < cw) a.x += q[0];
          if (c4 + 1 < cw) a.y += q[1];
          if (c4 + 2 < cw) a.z += q[2];
          if (c4 + 3 < cw) a.w += q[3];
        }
      }
      sh[c4 + 0][tap] = a.x; sh[c4 + 1][tap] = a.y; sh[c4 + 2][tap] = a.z; sh[c4 + 3][tap] = a.w;
    }
    __syncthreads();
    float* d = j.dst + ((long)n * j.Cin + c0) * j.taps;
    for (int idx = threadIdx.x; idx < cw * j.taps; idx += 512) {
      const int c = idx / j.taps, tap = idx - c * j.taps;
      const float v = j.scale * sh[c][tap];
      d[idx] = j.accumulate ? d[idx] + v : v;
    }
  }
}

// ---- channel-window elementwise ---------------------------------------------------
template <class T>
__global__ void lrelu_bwd_kernel(typename T::elem* g, int g_ctot, int g_off, const typename T::elem* y, int y_ctot,
                                 int y_off, int C, long nvox, float slope, const float* __restrict__ chan_scale,
                                 long vox_per_b) {
  const int c4 = C >> 2;
  const long total = nvox * c4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long v = i / c4;
    const int c = (int)(i % c4) * 4;
    typename T::elem* gp = g + v * g_ctot + g_off + c;
    float4 gv = ld4<T>(gp);
    const float4 yv = ld4<T>(y + v * y_ctot + y_off + c);
    float4 m = make_float4(yv.x > 0.f ? 1.f : slope, yv.y > 0.f ? 1.f : slope, yv.z > 0.f ? 1.f : slope,
                           yv.w > 0.f ? 1.f : slope);
    if (chan_scale) {  // Dropout3d keep factors [B][C]; y = lrelu(pre)*scale, so a dropped channel gets 0
      const float* cs = chan_scale + (v / vox_per_b) * C + c;
      m.x *= cs[0]; m.y *= cs[1]; m.z *= cs[2]; m.w *= cs[3];
    }
    gv.x *= m.x; gv.y *= m.y; gv.z *= m.z; gv.w *= m.w;
    st4<T>(gp, gv);
  }
}

template <class T>
__global__ void chan_axpby_kernel(typename T::elem* dst, int d_ctot, int d_off, const typename T::elem* src,
                                  int s_ctot, int s_off, int C, long nvox, float alpha, float beta) {
  const int c4 = C >> 2;
  const long total = nvox * c4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long v = i / c4;
    const int c = (int)(i % c4) * 4;
    typename T::elem* dp = dst + v * d_ctot + d_off + c;
    float4 s = ld4<T>(src + v * s_ctot + s_off + c);
    float4 o = make_float4(alpha * s.x, alpha * s.y, alpha * s.z, alpha * s.w);
    if (beta != 0.f) {
      const float4 d = ld4<T>(dp);
      o.x += beta * d.x;
      o.y += beta * d.y;
      o.z += beta * d.z;
      o.w += beta * d.w;
    }
    st4<T>(dp, o);
  }
}

template <class T>
__global__ void upsample2_bwd_kernel(const typename T::elem* dy, typename T::elem* dx, int B, int Xi, int Yi, int Zi,
                                     int C) {
  const int c4 = C >> 2;
  const long total = (long)B * Xi * Yi * Zi * c4;
  const long rowY = (long)Zi * C;        // one y step at the fine resolution
  const long rowX = (long)(2 * Yi) * rowY;  // one x step at the fine resolution
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4) * 4;
    long v = i / c4;
    const int z = (int)(v % Zi); v /= Zi;
    const int y = (int)(v % Yi); v /= Yi;
    const int x = (int)(v % Xi);
    const int b = (int)(v / Xi);
    const typename T::elem* p = dy + (((long)b * 2 * Xi + 2 * x) * 2 * Yi + 2 * y) * rowY + (long)z * C + c;
    const float4 a0 = ld4<T>(p), a1 = ld4<T>(p + rowY), a2 = ld4<T>(p + rowX), a3 = ld4<T>(p + rowX + rowY);
    st4<T>(dx + (i / c4) * C + c,
           make_float4(a0.x + a1.x + a2.x + a3.x, a0.y + a1.y + a2.y + a3.y, a0.z + a1.z + a2.z + a3.z,
                       a0.w + a1.w + a2.w + a3.w));
  }
}

// Sub-pixel filters of an up-sampling conv (see wsr_subpixel_fold): tap i of parity a collects the taps of the
// 3-wide filter that read the same un-sampled voxel:  a = 0: {0}, {1, 2};  a = 1: {0, 1}, {2}.
__device__ __forceinline__ void subpixel_set(int a, int i, int& lo, int& hi) {
  lo = a == 0 ? (i == 0 ? 0 : 1) : (i == 0 ? 0 : 2);
  hi = a == 0 ? (i == 0 ? 0 : 2) : (i == 0 ? 1 : 2);
}

__global__ void strided_parity_filters_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                                              int sz, int zc) {
  const int KZp = sz == 1 ? 3 : (zc == 0 ? 1 : 2);
  const long total = 4l * Cin * Cout * 4 * KZp;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int t = (int)(idx % KZp);
    long q = idx / KZp;
    const int j = (int)(q & 1), i = (int)((q >> 1) & 1);
    q >>= 2;
    const int co = (int)(q % Cout); q /= Cout;
    const int ci = (int)(q % Cin);
    const int ph = (int)(q / Cin), a = ph >> 1, b = ph & 1;
    const int kx = a == 0 ? (i == 0 ? 3 : 1) : (i == 0 ? 2 : 0);
    const int ky = b == 0 ? (j == 0 ? 3 : 1) : (j == 0 ? 2 : 0);
    const int kz = sz == 1 ? 2 - t : (zc == 0 ? 1 : (t == 0 ? 2 : 0));
    out[idx] = w[(((long)co * Cin + ci) * 16 + kx * 4 + ky) * 3 + kz];
  }
}

// Parity form of the FILTER gradient of a stride-(2,2,sz) (4,4,3) conv, padding 1: tap (kx, ky, kz) = (2i + a, 2j + b, .)
// only meets input voxels of one parity, so the gradient of the taps of class (a, b, zc) is a stride-1 (2,2,KZp)
// filter gradient over the input's sub-lattice (1 - a, 1 - b[, z class]).  This kernel moves the class gradients
// dwp (4, n = Cout*Cin, 2, 2, KZp) to their taps of the master gradient dw (n, 4, 4, 3): every master element is
// written by exactly one class.  sz = 1: kz = kk;  sz = 2: zc = 0 -> kz = 1, zc = 1 -> kz = 2 kk.
__global__ void strided_parity_unfold_kernel(const float* __restrict__ dwp, float* __restrict__ dw, long n, int sz, int zc) {
  const int KZp = sz == 1 ? 3 : (zc == 0 ? 1 : 2);
  const long total = 4 * n * 4 * KZp;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int kk = (int)(idx % KZp);
    long q = idx / KZp;
    const int j = (int)(q & 1), i = (int)((q >> 1) & 1);
    q >>= 2;
    const long f = q % n;
    const int ph = (int)(q / n), a = ph >> 1, b = ph & 1;
    const int kz = sz == 1 ? kk : (zc == 0 ? 1 : 2 * kk);
    dw[(f * 16 + (2 * i + a) * 4 + 2 * j + b) * 3 + kz] = dwp[idx];
  }
}

__global__ void subpixel_fold_kernel(const float* __restrict__ w, float* __restrict__ wp, long n, int KZ) {
  const long per = (long)4 * KZ;            // elements of one parity filter (2, 2, KZ)
  const long total = 4 * n * per;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int kz = (int)(idx % KZ);
    long q = idx / KZ;
    const int j = (int)(q & 1), i = (int)((q >> 1) & 1);
    q >>= 2;
    const long f = q % n;
    const int ph = (int)(q / n), a = ph >> 1, b = ph & 1;
    int x0, x1, y0, y1;
    subpixel_set(a, i, x0, x1);
    subpixel_set(b, j, y0, y1);
    float v = 0.f;
    for (int kx = x0; kx <= x1; ++kx)
      for (int ky = y0; ky <= y1; ++ky) v += w[(f * 9 + kx * 3 + ky) * KZ + kz];
    wp[idx] = v;
  }
}

__global__ void subpixel_unfold_kernel(const float* __restrict__ dwp, float* __restrict__ dw, long n, int KZ) {
  const long total = n * 9 * KZ;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int kz = (int)(idx % KZ);
    long q = idx / KZ;
    const int ky = (int)(q % 3), kx = (int)((q / 3) % 3);
    const long f = q / 9;
    float v = 0.f;
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) {
        const int i = a == 0 ? (kx == 0 ? 0 : 1) : (kx == 2 ? 1 : 0);  // the tap of parity a that holds kx
        const int j = b == 0 ? (ky == 0 ? 0 : 1) : (ky == 2 ? 1 : 0);
        v += dwp[((((long)(2 * a + b) * n + f) * 2 + i) * 2 + j) * KZ + kz];
      }
    dw[idx] = v;
  }
}

template <class T>
__global__ void planar_to_ndhwc_kernel(const float* __restrict__ src, typename T::elem* __restrict__ dst, int B, int C,
                                       long vpb, int d_ctot, int d_off, int cfill) {
  // writes channels [d_off, d_off + cfill): the first C from src, the rest zero
  const long total = (long)B * vpb * cfill;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % cfill);
    const long bv = i / cfill;
    const long v = bv % vpb;
    const int b = (int)(bv / vpb);
    const float x = c < C ? src[((long)b * C + c) * vpb + v] : 0.f;
    stf<T>(dst + bv * d_ctot + d_off + c, x);
  }
}

template <class T>
__global__ void ndhwc_to_planar_kernel(const typename T::elem* __restrict__ src, float* __restrict__ dst, int B, int C,
                                       long vpb, int s_ctot, int s_off) {
  const long total = (long)B * C * vpb;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long v = i % vpb;
    const long bc = i / vpb;
    const int c = (int)(bc % C);
    const int b = (int)(bc / C);
    dst[i] = ldf<T>(src + ((long)b * vpb + v) * s_ctot + s_off + c);
  }
}

// ---- BatchNorm3d ----------------------------------------------------------------------
// Threads stride over the flat [nvox][C] tensor with a stride that is a multiple
// of C, so each thread stays on one channel; per-workgroup LDS reduction over the
// threads sharing a channel, then one float atomic per channel per workgroup.
template <class T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const typename T::elem* __restrict__ x, int C, long nvox,
                                                      const float* __restrict__ shift_c, float* sums) {
  // sums of d = x - shift[c] and d^2.  Called twice per layer: shift = 0 gives the mean,
  // shift = mean gives a variance free of the E[x^2] - mean^2 cancellation.
  __shared__ float s1[256], s2[256];
  const long total = nvox * C;
  const long stride = (long)gridDim.x * 256;
  const float shift = shift_c ? shift_c[threadIdx.x % C] : 0.f;
  float a = 0.f, b = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += stride) {
    const float v = ldf<T>(x + i) - shift;
    a += v;
    b += v * v;
  }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = b;
  __syncthreads();
  if ((int)threadIdx.x < C) {  // C divides 256: threads t, t+C, t+2C.. share channel t
    for (int j = threadIdx.x + C; j < 256; j += C) {
      a += s1[j];
      b += s2[j];
    }
    atomicAdd(sums + threadIdx.x, a);
    atomicAdd(sums + C + threadIdx.x, b);
  }
}

// mean[c] = sums[c] / count
__global__ void bn_mean_kernel(const float* __restrict__ sums, const float* count_dev, float count_host,
                               float* __restrict__ mean, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float cnt = count_dev ? *count_dev : count_host;
  mean[c] = sums[c] / cnt;
}

// SyncBN, the per-channel algebra around its one collective (reference torch_blocks.py:20-25 under data parallelism;
// engine.py DiscriminatorProgram.forward): every rank sends its local mean and its local centred second moment
//   send[g][c] = mean_r ,  send[g][C + c] = M2_r = sum d^2 - (sum d)^2 / n      (d = x - mean_r; sum d is ~0, not 0)
// and combines the gathered shards by the pairwise rule for equal counts (Chan et al.), ranks added in index order:
//   mean = avg_r mean_r ,  M2 = sum_r M2_r + n * sum_r (mean_r - mean)^2
// written where bn_finalize reads them (work[g][c] = mean; s2[g][c] = 0, s2[g][C + c] = M2).  Was ~12 torch ops per layer.
__global__ void bn_shard_stats_kernel(const float* __restrict__ work, int work_stride, const float* __restrict__ s2,
                                      int s2_stride, float count, float* __restrict__ send, int G, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G * C) return;
  const int g = i / C, c = i - g * C;
  const float sd = s2[(long)g * s2_stride + c], sdd = s2[(long)g * s2_stride + C + c];
  send[(long)g * 2 * C + c] = work[(long)g * work_stride + c];
  send[(long)g * 2 * C + C + c] = sdd - sd * sd / count;
}

__global__ void bn_combine_shards_kernel(const float* __restrict__ allr, int world, float count, float* __restrict__ work,
                                         int work_stride, float* __restrict__ s2, int s2_stride, int G, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G * C) return;
  const int g = i / C, c = i - g * C;
  const long rs = (long)G * 2 * C;  // one rank's record
  const float* p = allr + (long)g * 2 * C + c;
  float msum = 0.f, m2 = 0.f;
  for (int r = 0; r < world; ++r) { msum += p[r * rs]; m2 += p[r * rs + C]; }
  const float mean = msum / (float)world;
  float dev = 0.f;
  for (int r = 0; r < world; ++r) { const float d = p[r * rs] - mean; dev += d * d; }
  work[(long)g * work_stride + c] = mean;
  s2[(long)g * s2_stride + c] = 0.f;
  s2[(long)g * s2_stride + C + c] = m2 + count * dev;
}

// from the shifted sums {sum d, sum d^2}, d = x - mean: biased variance, invstd, running-stat update
// (nn.BatchNorm3d: momentum update with the unbiased variance)
// per-channel tail of the two-pass statistics: s1 = sum d, s2 = sum d^2 with d = x - mean.  Every product and sum is rounded
// on its own (no contraction): the two kernels that call this (bn_finalize_kernel, bn_final_finalize_kernel) give the same
// bits whatever the compiler would have fused in either.
__device__ __forceinline__ float bn_finalize_channel(float s1, float s2, float cnt, float mean, float eps, float momentum,
                                                     float* invstd, float* running_mean, float* running_var) {
#pragma clang fp contract(off)  // (HIP's __fmul_rn / __fadd_rn are plain operators: they do not stop the contraction)
  const float dm = s1 / cnt;
  float var = s2 / cnt - dm * dm;
  var = var > 0.f ? var : 0.f;
  *invstd = rsqrtf(var + eps);
  if (running_mean) {
    const float unb = var * (cnt / fmaxf(cnt - 1.f, 1.f));
    const float keep = 1.f - momentum;
    *running_mean = keep * *running_mean + momentum * mean;
    *running_var = keep * *running_var + momentum * unb;
  }
  return var;
}

__global__ void bn_finalize_kernel(const float* __restrict__ sums2, const float* count_dev, float count_host,
                                   const float* __restrict__ mean, float eps, float momentum, float* invstd,
                                   float* var_out, float* running_mean, float* running_var, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float cnt = count_dev ? *count_dev : count_host;
  const float var = bn_finalize_channel(sums2[c], sums2[C + c], cnt, mean[c], eps, momentum, invstd + c,
                                        running_mean ? running_mean + c : nullptr, running_var ? running_var + c : nullptr);
  if (var_out) var_out[c] = var;
}

template <class T>
__global__ void bn_apply_kernel(const typename T::elem* __restrict__ x, typename T::elem* __restrict__ y,
                                const float* mean, const float* invstd, const float* gamma, const float* beta, int C,
                                long nvox, int act, float slope) {
  const long total = nvox * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    float v = (ldf<T>(x + i) - mean[c]) * invstd[c] * gamma[c] + beta[c];
    if (act) v = v > 0.f ? v : v * slope;
    stf<T>(y + i, v);
  }
}

template <class T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(typename T::elem* dy, const typename T::elem* __restrict__ y,
                                                           const typename T::elem* __restrict__ x, const float* mean,
                                                           const float* invstd, int C, long nvox, int act, float slope,
                                                           float* sums) {
  __shared__ float s1[256], s2[256];
  const long total = nvox * C;
  const long stride = (long)gridDim.x * 256;
  const int c = threadIdx.x % C;
  const float mu = mean[c], is = invstd[c];
  float a = 0.f, b = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += stride) {
    float g = ldf<T>(dy + i);
    if (act) {
      g *= ldf<T>(y + i) > 0.f ? 1.f : slope;
      stf<T>(dy + i, g);
    }
    const float xh = (ldf<T>(x + i) - mu) * is;
    a += g;
    b += g * xh;
  }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = b;
  __syncthreads();
  if ((int)threadIdx.x < C) {
    for (int j = threadIdx.x + C; j < 256; j += C) {
      a += s1[j];
      b += s2[j];
    }
    atomicAdd(sums + threadIdx.x, a);
    atomicAdd(sums + C + threadIdx.x, b);
  }
}

template <class T>
__global__ void bn_bwd_apply_kernel(const typename T::elem* __restrict__ g, const typename T::elem* __restrict__ x,
                                    typename T::elem* dx, const typename T::elem* __restrict__ act, float slope,
                                    const float* mean, const float* invstd, const float* gamma,
                                    const float* sums, float inv_n, int C, long nvox) {
  const long total = nvox * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    float v = ldf<T>(g + i);
    if (act) v *= ldf<T>(act + i) > 0.f ? 1.f : slope;  // LeakyReLU derivative of the layer's output (eval-mode pass)
    if (sums) {
      const float xh = (ldf<T>(x + i) - mean[c]) * invstd[c];
      v = v - sums[c] * inv_n - xh * sums[C + c] * inv_n;
    }
    stf<T>(dx + i, v * gamma[c] * invstd[c]);
  }
}

// The same, 8 bf16 per thread (C a multiple of 8), with the LeakyReLU derivative of the layer's OUTPUT `act` folded in
// (eval-mode BatchNorm in a pass that wants no parameter gradients - D inside a generator iteration: one pass over
// the gradient instead of lrelu_bwd + this).
__global__ void bn_bwd_apply_v8_kernel(const unsigned short* __restrict__ g, const unsigned short* __restrict__ x,
                                       unsigned short* dx, const unsigned short* __restrict__ act, float slope,
                                       const float* mean, const float* invstd, const float* gamma, const float* sums,
                                       float inv_n, int C, long nvox) {
  const long total8 = nvox * C / 8;
  const long i0 = blockIdx.x * (long)blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  // C divides 256 (checked on the host) and the grid stride is a multiple of 256 threads x 8 elements: a thread meets
  // the same 8 channels on every trip - their constants are fetched once (fetched per element, this pass was 2.2 x
  // slower than the scalar kernel it replaced)
  const int c0 = (int)((i0 * 8) % C);
  float sc[8], a0[8], a1[8], mu[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c0 + k;
    const float is = invstd[c];
    sc[k] = gamma[c] * is;
    mu[k] = mean[c];
    a0[k] = sums ? sums[c] * inv_n : 0.f;
    a1[k] = sums ? sums[C + c] * inv_n * is : 0.f;  // v -= (x - mean) * invstd * sum_gxhat / n
  }
  for (long i = i0; i < total8; i += stride) {
    const uint4 gv = *reinterpret_cast<const uint4*>(g + i * 8);
    uint4 av = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u), xv = make_uint4(0u, 0u, 0u, 0u);
    if (act) av = *reinterpret_cast<const uint4*>(act + i * 8);
    if (sums) xv = *reinterpret_cast<const uint4*>(x + i * 8);
    const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, aw[4] = {av.x, av.y, av.z, av.w}, xw[4] = {xv.x, xv.y, xv.z, xv.w};
    unsigned ow[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float r[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = 2 * k + h;
        const unsigned short gb = h ? (unsigned short)(gw[k] >> 16) : (unsigned short)(gw[k] & 0xFFFFu);
        const unsigned short ab = h ? (unsigned short)(aw[k] >> 16) : (unsigned short)(aw[k] & 0xFFFFu);
        const unsigned short xb = h ? (unsigned short)(xw[k] >> 16) : (unsigned short)(xw[k] & 0xFFFFu);
        float v = bf2f(gb) * ((short)ab > 0 ? 1.f : slope);  // bf16 sign test on the raw bits: y > 0
        v = v - a0[e] - (bf2f(xb) - mu[e]) * a1[e];
        r[h] = v * sc[e];
      }
      ow[k] = (unsigned)f2bf(r[0]) | ((unsigned)f2bf(r[1]) << 16);
    }
    *reinterpret_cast<uint4*>(dx + i * 8) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
  }
}

// ---- Adam --------------------------------------------------------------------------------
__global__ void adam_kernel(float* p, const float* __restrict__ g, float* m, float* v, long n, float step_size,
                            float beta1, float beta2, float eps, float wd, float bc2_sqrt) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i];
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

// one workgroup per job (a chunk of one tensor); the arithmetic of adam_kernel, element for element
struct AdamK { float step_size, beta1, omb1, beta2, omb2, eps, wd, bc2_sqrt; };
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamK& k) {
  if (k.wd != 0.f) g += k.wd * p;
  m = k.beta1 * m + k.omb1 * g;
  v = k.beta2 * v + k.omb2 * g * g;
  const float denom = sqrtf(v) / k.bc2_sqrt + k.eps;
  p = p - k.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const wsr_adam_job_t* __restrict__ jobs, const AdamK k) {
  const wsr_adam_job_t j = jobs[blockIdx.x];
  const long n = j.n;
  const bool vec = ((((size_t)j.p) | ((size_t)j.g) | ((size_t)j.m) | ((size_t)j.v)) & 15) == 0;
  long done = 0;
  if (vec) {
    const long n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(j.p);
    const float4* g4 = reinterpret_cast<const float4*>(j.g);
    float4* m4 = reinterpret_cast<float4*>(j.m);
    float4* v4 = reinterpret_cast<float4*>(j.v);
    for (long i = threadIdx.x; i < n4; i += 256) {
      float4 p = p4[i], m = m4[i], v = v4[i];
      const float4 g = g4[i];
      adam_one(p.x, g.x, m.x, v.x, k);
      adam_one(p.y, g.y, m.y, v.y, k);
      adam_one(p.z, g.z, m.z, v.z, k);
      adam_one(p.w, g.w, m.w, v.w, k);
      p4[i] = p; m4[i] = m; v4[i] = v;
    }
    done = n4 << 2;
  }
  for (long i = done + threadIdx.x; i < n; i += 256) {
    float p = j.p[i], m = j.m[i], v = j.v[i];
    adam_one(p, j.g[i], m, v, k);
    j.p[i] = p; j.m[i] = m; j.v[i] = v;
  }
}


// per-channel sum over voxels of an NDHWC window, pass 1: thread = (voxel lane, 4-channel group); the
// workgroup's partial sums go to row blockIdx.x of `part` (no atomics: C addresses would serialise them)
template <class T>
__global__ __launch_bounds__(256) void chan_sum_kernel(const typename T::elem* __restrict__ x, int ctot, int off, int C,
                                                      long nvox, float* __restrict__ part) {
  __shared__ float4 sh[256];
  const int groups = C >> 2;              // 4-channel groups (<= 256)
  const int lanes = 256 / groups;         // voxel lanes per workgroup
  const int g = threadIdx.x % groups, vl = threadIdx.x / groups;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (vl < lanes) {
    const long step = (long)gridDim.x * lanes;
    long v = (long)blockIdx.x * lanes + vl;
    for (; v + 7 * step < nvox; v += 8 * step) {  // eight independent loads in flight
      float4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = ld4<T>(x + (v + u * step) * ctot + off + 4 * g);
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w; }
    }
    for (; v < nvox; v += step) {
      const float4 a = ld4<T>(x + v * ctot + off + 4 * g);
      acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (vl == 0) {
    for (int l = 1; l < lanes; ++l) {
      const float4 a = sh[l * groups + g];
      acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
    *reinterpret_cast<float4*>(part + (long)blockIdx.x * C + 4 * g) = acc;
  }
}

// pass 2: out[c] = scale * sum over rows of part[row][c]; one workgroup, thread = (row lane, channel)
__global__ __launch_bounds__(1024) void chan_sum_final_kernel(const float* __restrict__ part, int rows, int C, float scale,
                                                             float* __restrict__ out) {
  __shared__ float sh[1024];
  const int lanes = 1024 / C;  // C <= 1024
  const int c = threadIdx.x % C, rl = threadIdx.x / C;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (rl < lanes) {
    int r = rl;
    for (; r + 3 * lanes < rows; r += 4 * lanes) {
      a0 += part[(long)r * C + c];
      a1 += part[(long)(r + lanes) * C + c];
      a2 += part[(long)(r + 2 * lanes) * C + c];
      a3 += part[(long)(r + 3 * lanes) * C + c];
    }
    for (; r < rows; r += lanes) a0 += part[(long)r * C + c];
  }
  sh[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rl == 0) {
    float a = sh[c];
    for (int l = 1; l < lanes; ++l) a += sh[l * C + c];
    out[c] = scale * a;
  }
}

// BatchNorm reductions, vectorised and atomic-free: thread = (voxel lane, 4-channel group) as in chan_sum;
// row blockIdx.x of `part` gets {sum a [C], sum b [C]} of the workgroup, chan_sum_final_kernel adds the rows.
//   stats : a = d, b = d*d, d = x - shift[c]
//   bwd   : g = dy * lrelu'(y) (written back when act), a = g, b = g * xhat
template <class T, bool BWD>
__global__ __launch_bounds__(256) void bn_reduce_v4_kernel(typename T::elem* dy, const typename T::elem* __restrict__ y,
                                                          const typename T::elem* __restrict__ x,
                                                          const float* __restrict__ mean_or_shift,
                                                          const float* __restrict__ invstd, int C, long nvox, int act,
                                                          float slope, float* __restrict__ part) {
  __shared__ float4 sa[256], sb[256];
  const int groups = C >> 2, lanes = 256 / groups;
  const int g = threadIdx.x % groups, vl = threadIdx.x / groups;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (!BWD && gridDim.y > 1) {
    // statistics of several BATCH GROUPS in one launch (wsr_bn_train_stats: D(real) and D(fake) of an iteration keep their
    // own statistics): group blockIdx.y owns voxels [y * nvox, (y + 1) * nvox), row 2C + C.. of the shift table, and the
    // partial rows [y * gridDim.x, ...)
    x += (long)blockIdx.y * nvox * C;
    if (mean_or_shift) mean_or_shift += (long)blockIdx.y * 2 * C;
    part += (long)blockIdx.y * gridDim.x * 2 * C;
  }
  if (vl < lanes) {
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = make_float4(1.f, 1.f, 1.f, 1.f);
    if (mean_or_shift) mu = *reinterpret_cast<const float4*>(mean_or_shift + 4 * g);
    if (BWD) is = *reinterpret_cast<const float4*>(invstd + 4 * g);
    const long step = (long)gridDim.x * lanes;
    for (long v = (long)blockIdx.x * lanes + vl; v < nvox; v += step) {
      const long o = v * C + 4 * g;
      const float4 xv = ld4<T>(x + o);
      if (BWD) {
        float4 gv = ld4<T>(dy + o);
        if (act) {
          const float4 yv = ld4<T>(y + o);
          gv.x *= yv.x > 0.f ? 1.f : slope; gv.y *= yv.y > 0.f ? 1.f : slope;
          gv.z *= yv.z > 0.f ? 1.f : slope; gv.w *= yv.w > 0.f ? 1.f : slope;
          st4<T>(dy + o, gv);
        }
        a.x += gv.x; a.y += gv.y; a.z += gv.z; a.w += gv.w;
        b.x += gv.x * (xv.x - mu.x) * is.x; b.y += gv.y * (xv.y - mu.y) * is.y;
        b.z += gv.z * (xv.z - mu.z) * is.z; b.w += gv.w * (xv.w - mu.w) * is.w;
      } else {
        const float dx = xv.x - mu.x, dy_ = xv.y - mu.y, dz = xv.z - mu.z, dw = xv.w - mu.w;
        a.x += dx; a.y += dy_; a.z += dz; a.w += dw;
        b.x += dx * dx; b.y += dy_ * dy_; b.z += dz * dz; b.w += dw * dw;
      }
    }
  }
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = b;
  __syncthreads();
  if (vl == 0) {
    for (int l = 1; l < lanes; ++l) {
      const float4 p = sa[l * groups + g], q = sb[l * groups + g];
      a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
      b.x += q.x; b.y += q.y; b.z += q.z; b.w += q.w;
    }
    float* row = part + (long)blockIdx.x * 2 * C;
    *reinterpret_cast<float4*>(row + 4 * g) = a;
    *reinterpret_cast<float4*>(row + C + 4 * g) = b;
  }
}

static inline int bn_v4_grid(int C, long nvox) {
  const int lanes = 256 / (C / 4);
  long grid = (nvox + (long)lanes * 16 - 1) / ((long)lanes * 16);
  if (grid > WSR_CHAN_SUM_ROWS) grid = WSR_CHAN_SUM_ROWS;
  return grid < 1 ? 1 : (int)grid;
}

// z-fold / z-unfold (see windsr_hip.h): planar tensors are (B, channels, planes, Z) with z contiguous
__global__ void zfold_kernel(const float* __restrict__ t, float* __restrict__ y, const float* __restrict__ bias, int B,
                             int C, int KZ, int pz, long planes, int Z) {
  const long total = (long)B * C * planes * Z;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int z = (int)(i % Z);
    const long r = i / Z;
    const long p = r % planes;
    const long bc = r / planes;
    const int c = (int)(bc % C);
    const long b = bc / C;
    float acc = bias ? bias[c] : 0.f;
    const float* tp = t + ((b * C + c) * KZ * planes + p) * Z;
    for (int kz = 0; kz < KZ; ++kz) {
      const int zz = z + kz - pz;
      if ((unsigned)zz < (unsigned)Z) acc += tp[(long)kz * planes * Z + zz];
    }
    y[i] = acc;
  }
}

template <class T>
__global__ void zunfold_kernel(const float* __restrict__ g, typename T::elem* __restrict__ d, int B, int C, int KZ, int pz,
                               long planes, int Z, int d_ctot, int d_off, int cfill) {
  const long total = (long)B * planes * Z * cfill;
  const int CK = C * KZ;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cfill);
    const long v = i / cfill;  // (b, p, z)
    float x = 0.f;
    if (ch < CK) {
      const int c = ch / KZ, kz = ch - c * KZ;
      const int z = (int)(v % Z);
      const long bp = v / Z;
      const long p = bp % planes, b = bp / planes;
      const int zz = z - kz + pz;
      if ((unsigned)zz < (unsigned)Z) x = g[((b * C + c) * planes + p) * Z + zz];
    }
    stf<T>(d + v * d_ctot + d_off + ch, x);
  }
}


// ---- y[b][n] = bias[n] + sum_k x[b][k] * w[n][k] for a handful of rows b and a very long k ---------------
// (the first classifier layer of the discriminator: 1..8 samples x 100 outputs x 131 072 features - a 52 MB
// weight matrix read once; the library GEMM picks a 13-workgroup tile for this shape and streams at 0.4 TB/s).
// One workgroup of 16 waves per output row: 64 KB of loads in flight per row, fp32 accumulation, block reduce.
template <int BMAX>
__global__ __launch_bounds__(1024) void linear_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int B,
                                                           int N, long K) {
  __shared__ float sh[BMAX][16];
  const int n = blockIdx.x;
  const float4* __restrict__ wr = reinterpret_cast<const float4*>(w + (long)n * K);
  float acc[BMAX];
#pragma unroll
  for (int b = 0; b < BMAX; ++b) acc[b] = 0.f;
  const long K4 = K >> 2;
  for (long i = threadIdx.x; i < K4; i += 1024) {
    const float4 wv = wr[i];
#pragma unroll
    for (int b = 0; b < BMAX; ++b)
      if (b < B) {
        const float4 xv = reinterpret_cast<const float4*>(x + (long)b * K)[i];
        acc[b] += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z + wv.w * xv.w;
      }
  }
#pragma unroll
  for (int b = 0; b < BMAX; ++b) {
    float a = acc[b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if ((threadIdx.x & 63) == 0) sh[b][threadIdx.x >> 6] = a;
  }
  __syncthreads();
  if (threadIdx.x < B) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) a += sh[threadIdx.x][k];  // fixed order: bit-reproducible
    y[(long)threadIdx.x * N + n] = a + (bias ? bias[n] : 0.f);
  }
}

// ---- per-channel sum of a planar fp32 tensor (B, C, V): the bias gradient of the last conv -----------------
// row blockIdx.y of `part` gets the C partial sums of slice blockIdx.y; chan_sum_final_kernel adds the rows
// (row = batch group x voxel slice: rv voxel slices per group of bper samples - a batch of 32 small patches used to
// leave this launch on 9 workgroups, 0.7 ms for 16 MB)
__global__ __launch_bounds__(256) void plane_sum_kernel(const float* __restrict__ src, int B, int C, long V,
                                                       float* __restrict__ part, int rv, int bper) {
  const int c = blockIdx.x;
  const int sl = blockIdx.y % rv, bg = blockIdx.y / rv;
  const long per = (V + rv - 1) / rv, v0 = (long)sl * per, v1 = v0 + per < V ? v0 + per : V;
  const int b0 = bg * bper, b1 = b0 + bper < B ? b0 + bper : B;
  float a = 0.f;
  for (int b = b0; b < b1; ++b) {
    const float* p = src + ((long)b * C + c) * V;
    for (long v = v0 + threadIdx.x; v < v1; v += 256) a += p[v];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[(long)blockIdx.y * C + c] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ---- wind-field derivatives (see windsr_hip.h); stencil rows: stencil.h
// forward: one thread per (b, comp, x, y, z) point, three derivatives
__global__ void wind_gradient_kernel(const float* __restrict__ f, const float* __restrict__ xs, const float* __restrict__ ys,
                                     const float* __restrict__ zc, float* __restrict__ out, int B, int X, int Y, int Z) {
  const long plane = (long)Y * Z, vol = (long)X * plane, total = (long)B * 3 * vol;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int z = (int)(i % Z);
    long r = i / Z;
    const int y = (int)(r % Y); r /= Y;
    const int x = (int)(r % X); r /= X;
    const int c = (int)(r % 3);
    const long b = r / 3;
    const float* fp = f + i;
    const float f0 = *fp;
    const long ob = ((b * 9) * vol) + (long)x * plane + (long)y * Z + z;
    {
      const Row3 w = deriv_row(Lin{xs, 1}, x, X);
      out[ob + (long)(0 + c) * vol] = (x > 0 ? w.a * fp[-plane] : 0.f) + w.b * f0 + (x < X - 1 ? w.c * fp[plane] : 0.f);
    }
    {
      const Row3 w = deriv_row(Lin{ys, 1}, y, Y);
      out[ob + (long)(3 + c) * vol] = (y > 0 ? w.a * fp[-Z] : 0.f) + w.b * f0 + (y < Y - 1 ? w.c * fp[Z] : 0.f);
    }
    {
      const float* zcol = zc + (b * vol + (long)x * plane + (long)y * Z);
      const Row3 w = deriv_row(Lin{zcol, 1}, z, Z);
      out[ob + (long)(6 + c) * vol] = (z > 0 ? w.a * fp[-1] : 0.f) + w.b * f0 + (z < Z - 1 ? w.c * fp[1] : 0.f);
    }
  }
}

// adjoint: df_j = c_{j-1} g_{j-1} + b_j g_j + a_{j+1} g_{j+1} along each axis, summed over the three axes
__global__ void wind_gradient_bwd_kernel(const float* __restrict__ g, const float* __restrict__ xs,
                                         const float* __restrict__ ys, const float* __restrict__ zc, float* __restrict__ df,
                                         int B, int X, int Y, int Z) {
  const long plane = (long)Y * Z, vol = (long)X * plane, total = (long)B * 3 * vol;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int z = (int)(i % Z);
    long r = i / Z;
    const int y = (int)(r % Y); r /= Y;
    const int x = (int)(r % X); r /= X;
    const int c = (int)(r % 3);
    const long b = r / 3;
    const long gb = ((b * 9) * vol) + (long)x * plane + (long)y * Z + z;
    float acc = 0.f;
    {
      const float* gp = g + gb + (long)(0 + c) * vol;
      const Lin co{xs, 1};
      acc += deriv_row(co, x, X).b * gp[0];
      if (x > 0) acc += deriv_row(co, x - 1, X).c * gp[-plane];
      if (x < X - 1) acc += deriv_row(co, x + 1, X).a * gp[plane];
    }
    {
      const float* gp = g + gb + (long)(3 + c) * vol;
      const Lin co{ys, 1};
      acc += deriv_row(co, y, Y).b * gp[0];
      if (y > 0) acc += deriv_row(co, y - 1, Y).c * gp[-Z];
      if (y < Y - 1) acc += deriv_row(co, y + 1, Y).a * gp[Z];
    }
    {
      const float* gp = g + gb + (long)(6 + c) * vol;
      const Lin co{zc + (b * vol + (long)x * plane + (long)y * Z), 1};
      acc += deriv_row(co, z, Z).b * gp[0];
      if (z > 0) acc += deriv_row(co, z - 1, Z).c * gp[-1];
      if (z < Z - 1) acc += deriv_row(co, z + 1, Z).a * gp[1];
    }
    df[i] = acc;
  }
}

}  // namespace

#define DISPATCH_T(dtype, CALL_BF16, CALL_F32) \
  do {                                         \
    if ((dtype) == WSR_BF16) {                 \
      CALL_BF16;                               \
    } else if ((dtype) == WSR_F32) {           \
      CALL_F32;                                \
    } else {                                   \
      return WSR_EINVAL;                       \
    }                                          \
  } while (0)

// ---- relativistic average GAN loss, forward + all partial derivatives (wsr_ragan_loss) --------------------------
// One workgroup: B is the per-GPU batch (1 .. 32 in every shipped configuration).  Sums are taken in a fixed order
// (lane-strided partials, then a shared-memory tree): bit-reproducible.
namespace {
__device__ __forceinline__ float rg_block_sum(float v, float* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (t < s) sh[t] += sh[t + s];
    __syncthreads();
  }
  const float r = sh[0];
  __syncthreads();
  return r;
}
__device__ __forceinline__ float rg_bce(float x, float t) {  // (1 - t) x - logsigmoid(x)
  return (1.f - t) * x - (fminf(x, 0.f) - log1pf(expf(-fabsf(x))));
}
__device__ __forceinline__ float rg_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void ragan_loss_kernel(const float* u, const float* v, const float* lu, const float* lv,
                                                         const float* mu_in, const float* mv_in, int B, float* out) {
  __shared__ float sh[256];
  const int t = threadIdx.x;
  float su = 0.f, sv = 0.f;
  if (!mu_in || !mv_in)
    for (int i = t; i < B; i += 256) { su += u[i]; sv += v[i]; }
  const float inv = 1.f / (float)B;
  const float mu = mu_in ? *mu_in : rg_block_sum(su, sh) * inv;
  const float mv = mv_in ? *mv_in : rg_block_sum(sv, sh) * inv;
  float loss = 0.f, sa = 0.f, sb = 0.f;
  for (int i = t; i < B; i += 256) {
    const float xu = u[i] - mv, xv = v[i] - mu;
    loss += rg_bce(xu, lu[i]) + rg_bce(xv, lv[i]);
    const float a = 0.5f * inv * (rg_sigmoid(xu) - lu[i]), b = 0.5f * inv * (rg_sigmoid(xv) - lv[i]);
    out[1 + i] = a;          // dL/d(u_i - mean v)
    out[1 + B + i] = b;      // dL/d(v_i - mean u)
    sa += a;
    sb += b;
  }
  loss = rg_block_sum(loss, sh);
  sa = rg_block_sum(sa, sh);
  sb = rg_block_sum(sb, sh);
  // dL/d mean(u) = -sum b, dL/d mean(v) = -sum a: handed to the caller, or folded into du / dv (mean over these B)
  if (!mu_in)
    for (int i = t; i < B; i += 256) out[1 + i] -= sb * inv;
  if (!mv_in)
    for (int i = t; i < B; i += 256) out[1 + B + i] -= sa * inv;
  if (t == 0) {
    out[0] = 0.5f * inv * loss;
    out[2 * B + 1] = mu_in ? -sb : 0.f;
    out[2 * B + 2] = mv_in ? -sa : 0.f;
  }
}
}  // namespace

extern "C" int wsr_ragan_loss(const float* u, const float* v, const float* lu, const float* lv, const float* mu, const float* mv,
                              int32_t B, float* out, void* stream) {
  if (!u || !v || !lu || !lv || !out || B < 1 || B > 65536) return WSR_EINVAL;
  hipLaunchKernelGGL(ragan_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), u, v, lu, lv, mu, mv, (int)B, out);
  WSR_LAUNCH_CHECK();
  return 0;
}

// ---- train-mode BatchNorm statistics of all batch groups of a layer in four launches (wsr_bn_train_stats) -------------
// The column sums of `rows` partial rows in chan_sum_final_kernel's order (bit-identical to the two-launch form).
namespace {
__device__ __forceinline__ float bn_rows_sum(const float* __restrict__ part, int rows, int C2, float* sh) {
  const int lanes = 1024 / C2;
  const int c = threadIdx.x % C2, rl = threadIdx.x / C2;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (rl < lanes) {
    int r = rl;
    for (; r + 3 * lanes < rows; r += 4 * lanes) {
      a0 += part[(long)r * C2 + c];
      a1 += part[(long)(r + lanes) * C2 + c];
      a2 += part[(long)(r + 2 * lanes) * C2 + c];
      a3 += part[(long)(r + 3 * lanes) * C2 + c];
    }
    for (; r < rows; r += lanes) a0 += part[(long)r * C2 + c];
  }
  __syncthreads();  // (sh is reused between calls)
  sh[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  float a = 0.f;
  if (rl == 0) {
    a = sh[c];
    for (int l = 1; l < lanes; ++l) a += sh[l * C2 + c];
  }
  return a;  // valid for threads with rl == 0: column c
}

// pass 1 tail: mean of group blockIdx.x -> work[g][0 .. C)
__global__ __launch_bounds__(1024) void bn_final_mean_kernel(const float* __restrict__ part, int rows, int C, float cnt,
                                                            float* __restrict__ work) {
  __shared__ float sh[1024];
  const int g = blockIdx.x;
  const float a = bn_rows_sum(part + (long)g * rows * 2 * C, rows, 2 * C, sh);
  if (threadIdx.x < C) work[(long)g * 2 * C + threadIdx.x] = a / cnt;  // (threads < 2C hold rl == 0; columns < C are sum d)
}

// pass 2 tail: one workgroup, the groups IN ORDER (each is a call of its own to the running statistics)
__global__ __launch_bounds__(1024) void bn_final_finalize_kernel(const float* __restrict__ part, int rows, int C, int groups,
                                                                float cnt, float eps, float momentum, float* __restrict__ work,
                                                                float* running_mean, float* running_var) {
  __shared__ float sh[1024];
  __shared__ float col[1024];
  for (int g = 0; g < groups; ++g) {
    const float a = bn_rows_sum(part + (long)g * rows * 2 * C, rows, 2 * C, sh);
    if (threadIdx.x < 2 * C) col[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < C) {
      const int c = threadIdx.x;
      float* wg = work + (long)g * 2 * C;
      bn_finalize_channel(col[c], col[C + c], cnt, wg[c], eps, momentum, wg + C + c, running_mean ? running_mean + c : nullptr,
                          running_var ? running_var + c : nullptr);
    }
    __syncthreads();
  }
}
}  // namespace

extern "C" int wsr_bn_train_stats(const void* x, int32_t C, int64_t nvox_g, int32_t groups, float eps, float momentum,
                                  float* work, float* running_mean, float* running_var, float* partials, int32_t dtype,
                                  void* stream) {
  if (!x || !work || !partials || C <= 0 || C % 4 || C > 512 || nvox_g <= 0 || groups < 1 || groups > 64) return WSR_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return WSR_EINVAL;
  const int grid = bn_v4_grid(C, nvox_g);  // (<= WSR_CHAN_SUM_ROWS rows PER GROUP: the per-group launches' partition, same bits)
  auto kb = bn_reduce_v4_kernel<BF16, false>;
  auto kf = bn_reduce_v4_kernel<F32, false>;
  hipStream_t st = as_stream(stream);
  const float cnt = (float)nvox_g;
  for (int pass = 0; pass < 2; ++pass) {
    const float* shift = pass ? work : nullptr;  // pass 2: d = x - mean of the group (work[g][0 .. C), row stride 2C)
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(kb, dim3(grid, groups), dim3(256), 0, st, (unsigned short*)nullptr, (const unsigned short*)nullptr,
                                  (const unsigned short*)x, shift, (const float*)nullptr, C, (long)nvox_g, 0, 0.f, partials),
               hipLaunchKernelGGL(kf, dim3(grid, groups), dim3(256), 0, st, (float*)nullptr, (const float*)nullptr,
                                  (const float*)x, shift, (const float*)nullptr, C, (long)nvox_g, 0, 0.f, partials));
    WSR_LAUNCH_CHECK();
    if (pass == 0)
      hipLaunchKernelGGL(bn_final_mean_kernel, dim3(groups), dim3(1024), 0, st, partials, grid, C, cnt, work);
    else
      hipLaunchKernelGGL(bn_final_finalize_kernel, dim3(1), dim3(1024), 0, st, partials, grid, C, groups, cnt, eps, momentum,
                         work, running_mean, running_var);
    WSR_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int wsr_abi_version(void) { return WSR_ABI_VERSION; }

extern "C" const char* wsr_error_string(int code) {
  switch (code) {
    case WSR_OK: return "ok";
    case WSR_EINVAL: return "invalid argument (geometry / null pointer)";
    case WSR_EUNSUPPORTED: return "shape not supported by the gfx950 kernels";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

extern "C" int wsr_pack_filter(const float* w, void* out, int32_t dtype, int32_t Cout, int32_t taps, int32_t Cin,
                               int32_t transpose, int32_t kpad, void* stream) {
  if (!w || !out || Cout <= 0 || taps <= 0 || Cin <= 0) return WSR_EINVAL;
  if (kpad < (transpose ? Cout : Cin)) return WSR_EINVAL;
  const long total = (long)(transpose ? Cin : Cout) * taps * kpad;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(pack_filter_kernel<BF16>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), w,
                                (unsigned short*)out, Cout, taps, Cin, transpose, kpad),
             hipLaunchKernelGGL(pack_filter_kernel<F32>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), w,
                                (float*)out, Cout, taps, Cin, transpose, kpad));
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_unpack_wgrad(const float* src, float* dst, int32_t Cout, int32_t taps, int32_t Cin, int32_t kpad,
                                float scale, int32_t accumulate, void* stream) {
  if (!src || !dst || Cout <= 0 || taps <= 0 || Cin <= 0 || kpad < Cin) return WSR_EINVAL;
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(ew_grid((long)Cout * Cin * taps)), dim3(EW_BLOCK), 0,
                     as_stream(stream), src, dst, Cout, taps, Cin, kpad, scale, accumulate);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_unpack_wgrad_multi(const wsr_unpack_job_t* jobs_dev, int32_t n_jobs, void* stream) {
  if (!jobs_dev || n_jobs <= 0 || n_jobs > 65535) return WSR_EINVAL;
  hipLaunchKernelGGL(unpack_wgrad_multi_kernel, dim3(32, (unsigned)n_jobs), dim3(256), 0, as_stream(stream), jobs_dev);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_unpack_wgrad_reduce_multi(const wsr_unpack_job_t* jobs_dev, int32_t n_jobs, void* stream) {
  if (!jobs_dev || n_jobs <= 0 || n_jobs > 65535) return WSR_EINVAL;
  int gx = WSR_ENV_INT("WSR_UNPACK_GRID", 128);  // (tuning aid: workgroups per job)
  gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
  hipLaunchKernelGGL(unpack_reduce_multi_kernel, dim3((unsigned)gx, (unsigned)n_jobs), dim3(512), 0, as_stream(stream), jobs_dev,
                     WSR_ENV_INT("WSR_UNPACK_ROWS", 1));
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_lrelu_bwd_inplace(void* g, int32_t g_ctot, int32_t g_off, const void* y, int32_t y_ctot,
                                     int32_t y_off, int32_t C, int64_t nvox, float slope, const float* chan_scale,
                                     int64_t vox_per_b, int32_t dtype, void* stream) {
  if (!g || !y || C <= 0 || nvox <= 0 || (chan_scale && vox_per_b <= 0)) return WSR_EINVAL;
  if ((C | g_ctot | g_off | y_ctot | y_off) & 3) return WSR_EUNSUPPORTED;
  const long total = nvox * (C >> 2);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(lrelu_bwd_kernel<BF16>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream),
                                (unsigned short*)g, g_ctot, g_off, (const unsigned short*)y, y_ctot, y_off, C,
                                (long)nvox, slope, chan_scale, (long)vox_per_b),
             hipLaunchKernelGGL(lrelu_bwd_kernel<F32>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream),
                                (float*)g, g_ctot, g_off, (const float*)y, y_ctot, y_off, C, (long)nvox, slope,
                                chan_scale, (long)vox_per_b));
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_chan_axpby(void* dst, int32_t d_ctot, int32_t d_off, const void* src, int32_t s_ctot, int32_t s_off,
                              int32_t C, int64_t nvox, float alpha, float beta, int32_t dtype, void* stream) {
  if (!dst || !src || C <= 0 || nvox <= 0) return WSR_EINVAL;
  if ((C | d_ctot | d_off | s_ctot | s_off) & 3) return WSR_EUNSUPPORTED;
  const long total = nvox * (C >> 2);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(chan_axpby_kernel<BF16>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream),
                                (unsigned short*)dst, d_ctot, d_off, (const unsigned short*)src, s_ctot, s_off, C,
                                (long)nvox, alpha, beta),
             hipLaunchKernelGGL(chan_axpby_kernel<F32>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream),
                                (float*)dst, d_ctot, d_off, (const float*)src, s_ctot, s_off, C, (long)nvox, alpha,
                                beta));
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_upsample2_bwd(const void* dy, void* dx, int32_t B, int32_t Xi, int32_t Yi, int32_t Zi, int32_t C,
                                 int32_t dtype, void* stream) {
  if (!dy || !dx || B <= 0 || Xi <= 0 || Yi <= 0 || Zi <= 0 || C <= 0) return WSR_EINVAL;
  if (C & 3) return WSR_EUNSUPPORTED;
  const long total = (long)B * Xi * Yi * Zi * (C >> 2);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(upsample2_bwd_kernel<BF16>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream),
                                (const unsigned short*)dy, (unsigned short*)dx, B, Xi, Yi, Zi, C),
             hipLaunchKernelGGL(upsample2_bwd_kernel<F32>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream),
                                (const float*)dy, (float*)dx, B, Xi, Yi, Zi, C));
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_strided_parity_filters(const float* w, float* out, int32_t Cout, int32_t Cin, int32_t sz, int32_t zc,
                                          void* stream) {
  if (!w || !out || Cout <= 0 || Cin <= 0 || (sz != 1 && sz != 2) || zc < 0 || zc >= sz) return WSR_EINVAL;
  const int KZp = sz == 1 ? 3 : (zc == 0 ? 1 : 2);
  hipLaunchKernelGGL(strided_parity_filters_kernel, dim3(ew_grid(16l * Cin * Cout * KZp)), dim3(EW_BLOCK), 0,
                     as_stream(stream), w, out, Cout, Cin, sz, zc);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_strided_parity_unfold(const float* dwp, float* dw, int64_t n, int32_t sz, int32_t zc, void* stream) {
  if (!dwp || !dw || n <= 0 || (sz != 1 && sz != 2) || zc < 0 || zc >= sz) return WSR_EINVAL;
  const int KZp = sz == 1 ? 3 : (zc == 0 ? 1 : 2);
  hipLaunchKernelGGL(strided_parity_unfold_kernel, dim3(ew_grid(16 * n * KZp)), dim3(EW_BLOCK), 0, as_stream(stream), dwp,
                     dw, (long)n, sz, zc);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_subpixel_fold(const float* w, float* wp, int64_t n, int32_t KZ, void* stream) {
  if (!w || !wp || n <= 0 || KZ <= 0) return WSR_EINVAL;
  hipLaunchKernelGGL(subpixel_fold_kernel, dim3(ew_grid(16 * n * KZ)), dim3(EW_BLOCK), 0, as_stream(stream), w, wp,
                     (long)n, KZ);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_subpixel_unfold(const float* dwp, float* dw, int64_t n, int32_t KZ, void* stream) {
  if (!dwp || !dw || n <= 0 || KZ <= 0) return WSR_EINVAL;
  hipLaunchKernelGGL(subpixel_unfold_kernel, dim3(ew_grid(9 * n * KZ)), dim3(EW_BLOCK), 0, as_stream(stream), dwp, dw,
                     (long)n, KZ);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_planar_to_ndhwc(const float* src, void* dst, int32_t B, int32_t C, int64_t vox_per_b, int32_t d_ctot,
                                   int32_t d_off, int32_t c_fill, int32_t dtype, void* stream) {
  if (!src || !dst || B <= 0 || C <= 0 || vox_per_b <= 0 || c_fill < C || d_off < 0 || d_off + c_fill > d_ctot)
    return WSR_EINVAL;
  const long total = (long)B * vox_per_b * c_fill;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(planar_to_ndhwc_kernel<BF16>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0,
                                as_stream(stream), src, (unsigned short*)dst, B, C, (long)vox_per_b, d_ctot, d_off,
                                c_fill),
             hipLaunchKernelGGL(planar_to_ndhwc_kernel<F32>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream),
                                src, (float*)dst, B, C, (long)vox_per_b, d_ctot, d_off, c_fill));
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_ndhwc_to_planar(const void* src, float* dst, int32_t B, int32_t C, int64_t vox_per_b, int32_t s_ctot,
                                   int32_t s_off, int32_t dtype, void* stream) {
  if (!src || !dst || B <= 0 || C <= 0 || vox_per_b <= 0 || s_off < 0 || s_off + C > s_ctot) return WSR_EINVAL;
  const long total = (long)B * C * vox_per_b;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(ndhwc_to_planar_kernel<BF16>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0,
                                as_stream(stream), (const unsigned short*)src, dst, B, C, (long)vox_per_b, s_ctot,
                                s_off),
             hipLaunchKernelGGL(ndhwc_to_planar_kernel<F32>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream),
                                (const float*)src, dst, B, C, (long)vox_per_b, s_ctot, s_off));
  WSR_LAUNCH_CHECK();
  return 0;
}

static inline int bn_grid(long total) {
  long g = (total + 256L * 64 - 1) / (256L * 64);  // >= 64 elements per thread
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int wsr_bn_stats(const void* x, int32_t C, int64_t nvox, const float* shift, float* sums, float* partials,
                            int32_t dtype, void* stream) {
  if (!x || !sums || C <= 0 || nvox <= 0) return WSR_EINVAL;
  if (partials && C % 4 == 0 && C <= 512) {  // vectorised two-pass form: sums are overwritten
    const int grid = bn_v4_grid(C, nvox);
    auto kb = bn_reduce_v4_kernel<BF16, false>;
    auto kf = bn_reduce_v4_kernel<F32, false>;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(kb, dim3(grid), dim3(256), 0, as_stream(stream), (unsigned short*)nullptr,
                                  (const unsigned short*)nullptr, (const unsigned short*)x, shift, (const float*)nullptr,
                                  C, (long)nvox, 0, 0.f, partials),
               hipLaunchKernelGGL(kf, dim3(grid), dim3(256), 0, as_stream(stream), (float*)nullptr,
                                  (const float*)nullptr, (const float*)x, shift, (const float*)nullptr, C, (long)nvox, 0,
                                  0.f, partials));
    WSR_LAUNCH_CHECK();
    hipLaunchKernelGGL(chan_sum_final_kernel, dim3(1), dim3(1024), 0, as_stream(stream), partials, grid, 2 * C, 1.f,
                       sums);
    WSR_LAUNCH_CHECK();
    return 0;
  }
  if (C > 256 || 256 % C) return WSR_EUNSUPPORTED;
  const int grid = bn_grid(nvox * C);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(bn_stats_kernel<BF16>, dim3(grid), dim3(256), 0, as_stream(stream),
                                (const unsigned short*)x, C, (long)nvox, shift, sums),
             hipLaunchKernelGGL(bn_stats_kernel<F32>, dim3(grid), dim3(256), 0, as_stream(stream), (const float*)x, C,
                                (long)nvox, shift, sums));
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_bn_mean(const float* sums, const float* count_dev, float count_host, float* mean, int32_t C,
                           void* stream) {
  if (!sums || !mean || C <= 0 || (!count_dev && !(count_host > 0.f))) return WSR_EINVAL;
  hipLaunchKernelGGL(bn_mean_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), sums, count_dev,
                     count_host, mean, C);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_bn_shard_stats(const float* work, int32_t work_stride, const float* s2, int32_t s2_stride, float count,
                                  float* send, int32_t G, int32_t C, void* stream) {
  if (!work || !s2 || !send || G <= 0 || C <= 0 || work_stride < C || s2_stride < 2 * C || !(count > 0.f)) return WSR_EINVAL;
  hipLaunchKernelGGL(bn_shard_stats_kernel, dim3((G * C + 255) / 256), dim3(256), 0, as_stream(stream), work, work_stride,
                     s2, s2_stride, count, send, G, C);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_bn_combine_shards(const float* gathered, int32_t world, float count, float* work, int32_t work_stride,
                                     float* s2, int32_t s2_stride, int32_t G, int32_t C, void* stream) {
  if (!gathered || !work || !s2 || world <= 0 || G <= 0 || C <= 0 || work_stride < C || s2_stride < 2 * C || !(count > 0.f))
    return WSR_EINVAL;
  hipLaunchKernelGGL(bn_combine_shards_kernel, dim3((G * C + 255) / 256), dim3(256), 0, as_stream(stream), gathered, world,
                     count, work, work_stride, s2, s2_stride, G, C);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_bn_finalize(const float* sums2, const float* count_dev, float count_host, const float* mean,
                               float eps, float momentum, float* invstd, float* var_out, float* running_mean,
                               float* running_var, int32_t C, void* stream) {
  if (!sums2 || !mean || !invstd || C <= 0 || (!count_dev && !(count_host > 0.f))) return WSR_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return WSR_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), sums2, count_dev,
                     count_host, mean, eps, momentum, invstd, var_out, running_mean, running_var, C);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_bn_apply_lrelu(const void* x, void* y, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, int32_t C, int64_t nvox, int32_t act, float slope, int32_t dtype,
                                  void* stream) {
  if (!x || !y || !mean || !invstd || !gamma || !beta || C <= 0 || nvox <= 0) return WSR_EINVAL;
  const long total = nvox * C;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(bn_apply_kernel<BF16>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream),
                                (const unsigned short*)x, (unsigned short*)y, mean, invstd, gamma, beta, C, (long)nvox,
                                act, slope),
             hipLaunchKernelGGL(bn_apply_kernel<F32>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream),
                                (const float*)x, (float*)y, mean, invstd, gamma, beta, C, (long)nvox, act, slope));
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_bn_bwd_reduce(void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                                 int32_t C, int64_t nvox, int32_t act, float slope, float* sums, float* partials,
                                 int32_t dtype, void* stream) {
  if (!dy || !y || !x || !mean || !invstd || !sums || C <= 0 || nvox <= 0) return WSR_EINVAL;
  if (partials && C % 4 == 0 && C <= 512) {  // vectorised two-pass form: sums are overwritten
    const int grid = bn_v4_grid(C, nvox);
    auto kb = bn_reduce_v4_kernel<BF16, true>;
    auto kf = bn_reduce_v4_kernel<F32, true>;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(kb, dim3(grid), dim3(256), 0, as_stream(stream), (unsigned short*)dy,
                                  (const unsigned short*)y, (const unsigned short*)x, mean, invstd, C, (long)nvox, act,
                                  slope, partials),
               hipLaunchKernelGGL(kf, dim3(grid), dim3(256), 0, as_stream(stream), (float*)dy, (const float*)y,
                                  (const float*)x, mean, invstd, C, (long)nvox, act, slope, partials));
    WSR_LAUNCH_CHECK();
    hipLaunchKernelGGL(chan_sum_final_kernel, dim3(1), dim3(1024), 0, as_stream(stream), partials, grid, 2 * C, 1.f,
                       sums);
    WSR_LAUNCH_CHECK();
    return 0;
  }
  if (C > 256 || 256 % C) return WSR_EUNSUPPORTED;
  const int grid = bn_grid(nvox * C);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(bn_bwd_reduce_kernel<BF16>, dim3(grid), dim3(256), 0, as_stream(stream),
                                (unsigned short*)dy, (const unsigned short*)y, (const unsigned short*)x, mean, invstd, C,
                                (long)nvox, act, slope, sums),
             hipLaunchKernelGGL(bn_bwd_reduce_kernel<F32>, dim3(grid), dim3(256), 0, as_stream(stream), (float*)dy,
                                (const float*)y, (const float*)x, mean, invstd, C, (long)nvox, act, slope, sums));
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_bn_bwd_apply(const void* g, const void* x, void* dx, const float* mean, const float* invstd,
                                const float* gamma, const float* sums, float inv_n, const void* act_y, float slope,
                                int32_t C, int64_t nvox, int32_t dtype, void* stream) {
  if (!g || !x || !dx || !mean || !invstd || !gamma || C <= 0 || nvox <= 0) return WSR_EINVAL;
  const long total = nvox * C;
  // (the 8-wide form pays a prologue of 32 constant loads per thread: large tensors only - on the discriminator's deep
  // layers, a few hundred thousand elements, the one-element kernel with 8 x the threads is faster: 12 against 20 us)
  if (dtype == WSR_BF16 && total >= (8l << 20) && C % 8 == 0 && 256 % C == 0 &&
      !(((size_t)g | (size_t)x | (size_t)dx | (size_t)act_y) & 15)) {
    hipLaunchKernelGGL(bn_bwd_apply_v8_kernel, dim3(ew_grid(total / 8)), dim3(EW_BLOCK), 0, as_stream(stream),
                       (const unsigned short*)g, (const unsigned short*)x, (unsigned short*)dx,
                       (const unsigned short*)act_y, slope, mean, invstd, gamma, sums, inv_n, C, (long)nvox);
    WSR_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(bn_bwd_apply_kernel<BF16>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream),
                                (const unsigned short*)g, (const unsigned short*)x, (unsigned short*)dx,
                                (const unsigned short*)act_y, slope, mean, invstd, gamma, sums, inv_n, C, (long)nvox),
             hipLaunchKernelGGL(bn_bwd_apply_kernel<F32>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream),
                                (const float*)g, (const float*)x, (float*)dx, (const float*)act_y, slope, mean, invstd,
                                gamma, sums, inv_n, C, (long)nvox));
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int32_t step, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step <= 0) return WSR_EINVAL;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, as_stream(stream), p, g, m, v, (long)n,
                     (float)(lr / bc1), beta1, beta2, eps, weight_decay, (float)sqrt(bc2));
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_adam_multi(const wsr_adam_job_t* jobs_dev, int32_t n_jobs, double lr, double beta1, double beta2, double eps,
                              double weight_decay, int32_t step, void* stream) {
  if (!jobs_dev || n_jobs <= 0 || step < 1) return WSR_EINVAL;
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  AdamK k;
  k.step_size = (float)(lr / bc1);
  k.beta1 = (float)beta1; k.omb1 = (float)(1.0 - beta1);
  k.beta2 = (float)beta2; k.omb2 = (float)(1.0 - beta2);
  k.eps = (float)eps; k.wd = (float)weight_decay; k.bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)n_jobs), dim3(256), 0, as_stream(stream), jobs_dev, k);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_chan_sum(const void* x, int32_t x_ctot, int32_t x_off, int32_t C, int64_t nvox, float scale,
                            float* out, float* partials, int32_t dtype, void* stream) {
  if (!x || !out || !partials || C <= 0 || nvox <= 0 || x_off < 0 || x_off + C > x_ctot) return WSR_EINVAL;
  if (C % 4 || C > 1024 || x_ctot % 4 || x_off % 4) return WSR_EUNSUPPORTED;
  const int groups = C / 4, lanes = 256 / groups;
  long grid = (nvox + (long)lanes * 16 - 1) / ((long)lanes * 16);  // >= 16 voxels per thread
  if (grid > WSR_CHAN_SUM_ROWS) grid = WSR_CHAN_SUM_ROWS;
  if (grid < 1) grid = 1;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(chan_sum_kernel<BF16>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream),
                                (const unsigned short*)x, x_ctot, x_off, C, (long)nvox, partials),
             hipLaunchKernelGGL(chan_sum_kernel<F32>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream),
                                (const float*)x, x_ctot, x_off, C, (long)nvox, partials));
  WSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(chan_sum_final_kernel, dim3(1), dim3(1024), 0, as_stream(stream), partials, (int)grid, C, scale,
                     out);
  WSR_LAUNCH_CHECK();
  return 0;
}

static long chan_sum_grid(int C, long nvox) {
  const int groups = C / 4, lanes = 256 / groups;
  long grid = (nvox + (long)lanes * 16 - 1) / ((long)lanes * 16);  // >= 16 voxels per thread
  if (grid > WSR_CHAN_SUM_ROWS) grid = WSR_CHAN_SUM_ROWS;
  return grid < 1 ? 1 : grid;
}

extern "C" int wsr_chan_sum_rows(int32_t C, int64_t nvox) {
  if (C <= 0 || C % 4 || C > 1024 || nvox <= 0) return 0;
  return (int)chan_sum_grid(C, (long)nvox);
}

extern "C" int wsr_chan_sum_partials(const void* x, int32_t x_ctot, int32_t x_off, int32_t C, int64_t nvox,
                                     float* partials, int32_t dtype, void* stream) {
  if (!x || !partials || C <= 0 || nvox <= 0 || x_off < 0 || x_off + C > x_ctot) return WSR_EINVAL;
  if (C % 4 || C > 1024 || x_ctot % 4 || x_off % 4) return WSR_EUNSUPPORTED;
  const long grid = chan_sum_grid(C, (long)nvox);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(chan_sum_kernel<BF16>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream),
                                (const unsigned short*)x, x_ctot, x_off, C, (long)nvox, partials),
             hipLaunchKernelGGL(chan_sum_kernel<F32>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream),
                                (const float*)x, x_ctot, x_off, C, (long)nvox, partials));
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_linear_rows(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t N,
                               int64_t K, void* stream) {
  if (!x || !w || !y || B <= 0 || N <= 0 || K <= 0) return WSR_EINVAL;
  if (B > 8 || (K & 3) || (((size_t)x | (size_t)w) & 15)) return WSR_EUNSUPPORTED;
  if (B <= 2)
    hipLaunchKernelGGL(linear_rows_kernel<2>, dim3((unsigned)N), dim3(1024), 0, as_stream(stream), x, w, bias, y, B, N, (long)K);
  else
    hipLaunchKernelGGL(linear_rows_kernel<8>, dim3((unsigned)N), dim3(1024), 0, as_stream(stream), x, w, bias, y, B, N, (long)K);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_plane_sum(const float* src, int32_t B, int32_t C, int64_t V, float* out, float* partials, void* stream) {
  if (!src || !out || !partials || B <= 0 || C <= 0 || C > 1024 || V <= 0) return WSR_EINVAL;
  long rv = (V + 16383) / 16384;  // >= 64 elements per thread
  if (rv > WSR_CHAN_SUM_ROWS) rv = WSR_CHAN_SUM_ROWS;
  if (rv < 1) rv = 1;
  long groups = WSR_CHAN_SUM_ROWS / rv;  // batch groups: as many as the partial-sum rows allow
  if (groups > B) groups = B;
  if (groups < 1) groups = 1;
  const int bper = (int)((B + groups - 1) / groups);
  groups = (B + bper - 1) / bper;
  const long rows = rv * groups;
  hipLaunchKernelGGL(plane_sum_kernel, dim3((unsigned)C, (unsigned)rows), dim3(256), 0, as_stream(stream), src, B, C,
                     (long)V, partials, (int)rv, bper);
  WSR_LAUNCH_CHECK();
  hipLaunchKernelGGL(chan_sum_final_kernel, dim3(1), dim3(1024), 0, as_stream(stream), partials, (int)rows, C, 1.f, out);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_zfold(const float* t, float* y, const float* bias, int32_t B, int32_t C, int32_t KZ, int32_t pz,
                         int64_t planes, int32_t Z, void* stream) {
  if (!t || !y || B <= 0 || C <= 0 || KZ <= 0 || pz < 0 || planes <= 0 || Z <= 0) return WSR_EINVAL;
  const long total = (long)B * C * planes * Z;
  hipLaunchKernelGGL(zfold_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), t, y, bias, B, C, KZ,
                     pz, (long)planes, Z);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_zunfold(const float* g, void* d, int32_t B, int32_t C, int32_t KZ, int32_t pz, int64_t planes,
                           int32_t Z, int32_t d_ctot, int32_t d_off, int32_t c_fill, int32_t dtype, void* stream) {
  if (!g || !d || B <= 0 || C <= 0 || KZ <= 0 || pz < 0 || planes <= 0 || Z <= 0) return WSR_EINVAL;
  if (c_fill < C * KZ || d_off < 0 || d_off + c_fill > d_ctot) return WSR_EINVAL;
  const long total = (long)B * planes * Z * c_fill;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(zunfold_kernel<BF16>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), g,
                                (unsigned short*)d, B, C, KZ, pz, (long)planes, Z, d_ctot, d_off, c_fill),
             hipLaunchKernelGGL(zunfold_kernel<F32>, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), g,
                                (float*)d, B, C, KZ, pz, (long)planes, Z, d_ctot, d_off, c_fill));
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_wind_gradient(const float* f, const float* xs, const float* ys, const float* zc, float* out,
                                 int32_t B, int32_t X, int32_t Y, int32_t Z, void* stream) {
  if (!f || !xs || !ys || !zc || !out || B <= 0 || X <= 0 || Y <= 0 || Z <= 0) return WSR_EINVAL;
  const long total = (long)B * 3 * X * Y * Z;
  hipLaunchKernelGGL(wind_gradient_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), f, xs, ys, zc,
                     out, B, X, Y, Z);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_wind_gradient_bwd(const float* g, const float* xs, const float* ys, const float* zc, float* df,
                                     int32_t B, int32_t X, int32_t Y, int32_t Z, void* stream) {
  if (!g || !xs || !ys || !zc || !df || B <= 0 || X <= 0 || Y <= 0 || Z <= 0) return WSR_EINVAL;
  const long total = (long)B * 3 * X * Y * Z;
  hipLaunchKernelGGL(wind_gradient_bwd_kernel, dim3(ew_grid(total)), dim3(EW_BLOCK), 0, as_stream(stream), g, xs, ys, zc,
                     df, B, X, Y, Z);
  WSR_LAUNCH_CHECK();
  return 0;
}
