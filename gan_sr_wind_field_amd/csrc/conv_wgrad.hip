// Filter gradient of the 3-D convolution as a split-K GEMM on the MFMA 16x16 family.
//
//   dw[n, tap, c] += sum_m dy[m, n] * x[src(m, tap), c]
//
// One workgroup owns (tap, 16*TN output channels, 16*TC input channels, a
// contiguous range of output voxels).  Per K-step the dy rows and the gathered x
// rows of KB voxels are staged in LDS as [voxel][channel]; the reduction index
// (voxel) is the slow dimension of both operands, so bf16 fragments are fetched
// with the transposing LDS read (ds_read_b64_tr_b16) and fp32 fragments with
// plain ds_read_b32.  The four waves split the KB voxels of a step between them
// (wavefront-level partial sums), and each wave adds its tile into the fp32
// filter gradient with global float atomics once, at the very end.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef s16x4_t __attribute__((address_space(3))) * lds_s16x4_ptr;

struct WgradArgs {
  const char* x;
  const char* dy;
  float* dw;
  int B, Xi, Yi, Zi, Xo, Yo, Zo;
  int Cin, in_ctot, in_off;
  int Cout, out_ctot, out_off;
  int KX, KY, KZ, sx, sy, sz, px, py, pz, ups;
  int M;        // B*Xo*Yo*Zo
  int mchunk;   // voxels per workgroup (multiple of KB)
  int ntiles, ctiles;
  long part_stride;  // > 0: wave w of voxel range mc STORES its sums at dw + (4*mc + w)*part_stride (no atomics)
  int mchunks;       // voxel ranges (host side)
};

template <class T> struct WgCfg;
template <> struct WgCfg<BF16> { static constexpr int KB = 128; };  // 32 voxels / wave / step
template <> struct WgCfg<F32> { static constexpr int KB = 64; };    // 16 voxels / wave / step

template <class T, int TN, int TC>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
  using elem = typename T::elem;
  constexpr int ESZ = sizeof(elem);
  constexpr int KB = WgCfg<T>::KB;
  constexpr int BN = TN * 16, BC = TC * 16;
  constexpr int YS = BN * ESZ + 16;  // LDS row strides (bytes)
  constexpr int XS = BC * ESZ + 16;
  constexpr int YP = BN * ESZ / 16;  // pieces per dy row
  constexpr int XP = BC * ESZ / 16;  // pieces per x row
  constexpr int TPR = 256 / KB;      // threads per row (2 for bf16, 4 for fp32)
  constexpr int YPT = (YP + TPR - 1) / TPR;
  constexpr int XPT = (XP + TPR - 1) / TPR;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Yt = smem;            // [KB][YS]
  char* Xt = smem + KB * YS;  // [KB][XS]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 15, fg = lane >> 4;

  int bid = blockIdx.x;
  const int ct = bid % a.ctiles; bid /= a.ctiles;
  const int nt = bid % a.ntiles; bid /= a.ntiles;
  const int taps = a.KX * a.KY * a.KZ;
  const int tap = bid % taps;
  const int mc = bid / taps;
  const int kz = tap % a.KZ, ky = (tap / a.KZ) % a.KY, kx = tap / (a.KZ * a.KY);
  const int n0 = nt * BN, c0 = ct * BC;
  const int m_begin = mc * a.mchunk;
  const int m_end = min(a.M, m_begin + a.mchunk);

  const int row = t / TPR, sub = t % TPR;
  uint4 yr[YPT], xr[XPT];

  auto load_step = [&](int mbase) {
    const int m = mbase + row;
    bool ok = m < m_end;
    long xvox = 0;
    if (ok) {
      int zo = m % a.Zo;
      int q = m / a.Zo;
      int yo = q % a.Yo;
      q /= a.Yo;
      int xo = q % a.Xo;
      int b = q / a.Xo;
      int x = xo * a.sx - a.px + kx, y = yo * a.sy - a.py + ky, z = zo * a.sz - a.pz + kz;
      bool in = true;
      if (a.ups) {
        in = ((unsigned)x < (unsigned)(2 * a.Xi)) && ((unsigned)y < (unsigned)(2 * a.Yi));
        x >>= 1;
        y >>= 1;
      }
      in = in && ((unsigned)x < (unsigned)a.Xi) && ((unsigned)y < (unsigned)a.Yi) && ((unsigned)z < (unsigned)a.Zi);
      xvox = in ? (((long)b * a.Xi + x) * a.Yi + y) * a.Zi + z : -1;
    }
#pragma unroll
    for (int i = 0; i < YPT; ++i) {
      const int p = sub * YPT + i;  // piece inside the dy row
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok && p < YP && n0 + p * T::EPP < a.Cout) {
        const elem* src = reinterpret_cast<const elem*>(a.dy) + (long)m * a.out_ctot + a.out_off + n0 + p * T::EPP;
        if (n0 + (p + 1) * T::EPP <= a.Cout) {
          v = *reinterpret_cast<const uint4*>(src);
        } else {  // ragged tail of the channel window
          elem tmp[T::EPP];
#pragma unroll
          for (int e = 0; e < T::EPP; ++e) tmp[e] = (n0 + p * T::EPP + e < a.Cout) ? src[e] : (elem)0;
          v = *reinterpret_cast<uint4*>(tmp);
        }
      }
      yr[i] = v;
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int p = sub * XPT + i;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok && xvox >= 0 && p < XP && c0 + (p + 1) * T::EPP <= a.Cin) {
        const elem* src = reinterpret_cast<const elem*>(a.x) + xvox * a.in_ctot + a.in_off + c0 + p * T::EPP;
        v = *reinterpret_cast<const uint4*>(src);
      }
      xr[i] = v;
    }
  };

  f32x4_t acc[TN][TC];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TC; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if (m_begin < m_end) load_step(m_begin);
  for (int mbase = m_begin; mbase < m_end; mbase += KB) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < YPT; ++i) {
      const int p = sub * YPT + i;
      if (p < YP) *reinterpret_cast<uint4*>(Yt + row * YS + p * 16) = yr[i];
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int p = sub * XPT + i;
      if (p < XP) *reinterpret_cast<uint4*>(Xt + row * XS + p * 16) = xr[i];
    }
    __syncthreads();
    if (mbase + KB < m_end) load_step(mbase + KB);

    if constexpr (T::ID == WSR_BF16) {
      // this wave's 32 voxels: rows wave*32 + [0, 32)
      // tr-read: lane 4q+p of 16-lane group G supplies &tile[R0 + q][C0 + 4p]; it receives
      // column (lane&15) of rows R0..R0+3.  A/B fragment k = 8G + j  ->  R0 = 8G (+4).
      const int q4 = fr >> 2, p4 = fr & 3;
      const int r0 = wave * 32 + 8 * fg + q4;
      uint4 af[TN], bf[TC];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const char* base = Yt + r0 * YS + (i * 16 + 4 * p4) * 2;
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + 4 * YS));
        uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        af[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
      }
#pragma unroll
      for (int j = 0; j < TC; ++j) {
        const char* base = Xt + r0 * XS + (j * 16 + 4 * p4) * 2;
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + 4 * XS));
        uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        bf[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
      }
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[i]),
                                                              __builtin_bit_cast(bf16x8_t, bf[j]), acc[i][j], 0, 0, 0);
    } else {
      // this wave's 16 voxels, four K=4 sub-steps; A[n = fr][k = fg], B[k = fg][c = fr]
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int r = wave * 16 + ks * 4 + fg;
        float af[TN], bf[TC];
#pragma unroll
        for (int i = 0; i < TN; ++i) af[i] = *reinterpret_cast<const float*>(Yt + r * YS + (i * 16 + fr) * 4);
#pragma unroll
        for (int j = 0; j < TC; ++j) bf[j] = *reinterpret_cast<const float*>(Xt + r * XS + (j * 16 + fr) * 4);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // acc[i][j][reg]: n = n0 + i*16 + 4*fg + reg ; c = c0 + j*16 + fr
  float* const dwp = a.dw + (long)(4 * mc + wave) * a.part_stride;
  const bool store = a.part_stride > 0;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TC; ++j) {
      const int c = c0 + j * 16 + fr;
      if (c >= a.Cin) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + i * 16 + 4 * fg + r;
        if (n < a.Cout) {
          float* q = dwp + ((long)n * taps + tap) * a.Cin + c;
          if (store) *q = acc[i][j][r];
          else atomicAdd(q, acc[i][j][r]);
        }
      }
    }
}

// plan != nullptr: report the number of partial copies (4 per voxel range: one per wave) and do not launch
template <class T, int TN, int TC>
int launch_wgrad(WgradArgs& a, hipStream_t st, int* plan, int n_parts) {
  constexpr int KB = WgCfg<T>::KB;
  constexpr int ESZ = sizeof(typename T::elem);
  constexpr int BN = TN * 16, BC = TC * 16;
  a.ntiles = (a.Cout + BN - 1) / BN;
  a.ctiles = (a.Cin + BC - 1) / BC;
  const int taps = a.KX * a.KY * a.KZ;
  const long tiles = (long)taps * a.ntiles * a.ctiles;
  // aim for ~4096 workgroups; each owns a voxel range that is a multiple of KB
  long want = (4096 + tiles - 1) / tiles;
  if (want < 1) want = 1;
  long mchunk = ((a.M + want - 1) / want + KB - 1) / KB * KB;
  if (mchunk < 4 * KB) mchunk = 4 * KB;
  a.mchunk = (int)mchunk;
  const long mchunks = (a.M + mchunk - 1) / mchunk;
  if (plan) { *plan = (int)(4 * mchunks); return 0; }
  if (a.part_stride > 0 && n_parts != 4 * mchunks) return WSR_EINVAL;
  const size_t lds = (size_t)KB * ((BN + BC) * ESZ + 32);
  dim3 grid((unsigned)(tiles * mchunks)), block(256);
  hipLaunchKernelGGL((wgrad_kernel<T, TN, TC>), grid, block, lds, st, a);
  WSR_LAUNCH_CHECK();
  return 0;
}

template <class T>
int dispatch_wgrad(WgradArgs& a, hipStream_t st, int* plan, int n_parts) {
  if (a.Cout % 48 == 0 && a.Cin % 48 == 0 && a.Cin % 64 != 0) return launch_wgrad<T, 3, 3>(a, st, plan, n_parts);  // 144 x 144
  if (a.Cout <= 16)
    return a.Cin <= 16 ? launch_wgrad<T, 1, 1>(a, st, plan, n_parts) : launch_wgrad<T, 1, 4>(a, st, plan, n_parts);
  if (a.Cin <= 16) return launch_wgrad<T, 2, 1>(a, st, plan, n_parts);
  if (a.Cin <= 32) return launch_wgrad<T, 2, 2>(a, st, plan, n_parts);
  return launch_wgrad<T, 2, 4>(a, st, plan, n_parts);
}

}  // namespace

int wsr_wgrad_tile_bf16(const wsr_conv_t* c, const void* x, const void* dy, float* dw, int tri_base, int tri_step,
                        long part_stride, int n_parts, int* plan, void* stream, const void* x2, int x2_ctot,
                        int x2_c0);  // conv_wgrad_tile.hip

int wsr_wgrad_tile_f32(const wsr_conv_t* c, const void* x, const void* dy, float* dw, long part_stride, int n_parts,
                       int* plan, void* stream, const void* x2, int x2_ctot, int x2_c0);  // conv_wgrad_tile_f32.hip

// shared body: accumulate (part_stride = 0), deterministic parts (part_stride > 0) or plan only (plan != nullptr)
// x2 != NULL: the input's channels >= x2_c0 live in a second tensor (tile kernels only)
static int wgrad_any(const wsr_conv_t* c, const void* x, const void* dy, float* dw, int tri_base, int tri_step,
                     long part_stride, int n_parts, int* plan, void* stream, const void* x2 = nullptr, int x2_ctot = 0,
                     int x2_c0 = 0) {
  {  // stride-1 bf16 convs: LDS-tile kernel (x / dy read once per tile, not per tap)
    const int rc = wsr_wgrad_tile_bf16(c, x, dy, dw, tri_base, tri_step, part_stride, n_parts, plan, stream, x2, x2_ctot,
                                       x2_c0);
    if (rc != WSR_EUNSUPPORTED || tri_step > 0) return rc;
  }
  if (c->lat) return WSR_EUNSUPPORTED;  // parity convs of the sub-pixel form: tile kernels only
  if (c->dtype == WSR_F32 && tri_step == 0) {  // stride-1 fp32 convs: LDS-tile kernel (all taps per workgroup)
    const int rc = wsr_wgrad_tile_f32(c, x, dy, dw, part_stride, n_parts, plan, stream, x2, x2_ctot, x2_c0);
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  if (x2) return WSR_EUNSUPPORTED;  // the per-tap kernel reads one tensor
  const int epp = c->dtype == WSR_BF16 ? 8 : 4;
  if (c->Cin % epp || c->in_ctot % epp || c->in_off % epp) return WSR_EUNSUPPORTED;
  if (c->out_ctot % epp || c->out_off % epp) return WSR_EUNSUPPORTED;
  if ((long)c->B * c->Xo * c->Yo * c->Zo > 0x7fffffffL) return WSR_EUNSUPPORTED;
  WgradArgs a{};
  a.x = (const char*)x;
  a.dy = (const char*)dy;
  a.dw = dw;
  a.B = c->B; a.Xi = c->Xi; a.Yi = c->Yi; a.Zi = c->Zi;
  a.Xo = c->Xo; a.Yo = c->Yo; a.Zo = c->Zo;
  a.Cin = c->Cin; a.in_ctot = c->in_ctot; a.in_off = c->in_off;
  a.Cout = c->Cout; a.out_ctot = c->out_ctot; a.out_off = c->out_off;
  a.KX = c->KX; a.KY = c->KY; a.KZ = c->KZ;
  a.sx = c->sx; a.sy = c->sy; a.sz = c->sz;
  a.px = c->px; a.py = c->py; a.pz = c->pz;
  a.ups = c->upsample_xy ? 1 : 0;
  a.M = c->B * c->Xo * c->Yo * c->Zo;
  a.part_stride = part_stride;
  return c->dtype == WSR_BF16 ? dispatch_wgrad<BF16>(a, as_stream(stream), plan, n_parts)
                              : dispatch_wgrad<F32>(a, as_stream(stream), plan, n_parts);
}

// wsr_conv_split_ok's third leg (conv_tile.hip): would the FILTER gradient of this conv run on a tile kernel with its
// input in two tensors (channels >= c0 in the second)?  The planners are asked, nothing is launched; the second
// tensor's pointer is a non-NULL placeholder no plan path reads through.
int wsr_wgrad_split_plan_ok(const wsr_conv_t* c, int c0) {
  int plan = 0;
  static const char placeholder[16] = {0};
  return wgrad_any(c, nullptr, nullptr, nullptr, 0, 0, 0, 0, &plan, nullptr, placeholder, c->Cin - c0, c0) == 0 && plan > 0;
}

extern "C" int wsr_conv3d_wgrad_tri(const wsr_conv_t* c, const void* x, const void* dy, float* dw, int32_t tri_base,
                                    int32_t tri_step, void* stream) {
  if (!conv_geom_ok(c) || !x || !dy || !dw || tri_base <= 0 || tri_step <= 0) return WSR_EINVAL;
  return wgrad_any(c, x, dy, dw, tri_base, tri_step, 0, 0, nullptr, stream);
}

extern "C" int wsr_conv3d_wgrad(const wsr_conv_t* c, const void* x, const void* dy, float* dw, void* stream) {
  if (!conv_geom_ok(c) || !x || !dy || !dw) return WSR_EINVAL;
  return wgrad_any(c, x, dy, dw, 0, 0, 0, 0, nullptr, stream);
}

extern "C" int wsr_conv3d_wgrad_nparts(const wsr_conv_t* c, int32_t tri_base, int32_t tri_step, int32_t* n_parts) {
  if (!conv_geom_ok(c) || !n_parts || tri_base < 0 || tri_step < 0) return WSR_EINVAL;
  int plan = 0;
  const int rc = wgrad_any(c, nullptr, nullptr, nullptr, tri_base, tri_step, 0, 0, &plan, nullptr);
  *n_parts = plan;
  return rc;
}

extern "C" int wsr_conv3d_wgrad_parts(const wsr_conv_t* c, const void* x, const void* dy, float* parts,
                                      int64_t part_stride, int32_t n_parts, int32_t tri_base, int32_t tri_step,
                                      void* stream) {
  if (!conv_geom_ok(c) || !x || !dy || !parts || tri_base < 0 || tri_step < 0 || n_parts <= 0) return WSR_EINVAL;
  if (part_stride < (int64_t)c->Cout * c->KX * c->KY * c->KZ * c->Cin) return WSR_EINVAL;
  return wgrad_any(c, x, dy, parts, tri_base, tri_step, (long)part_stride, n_parts, nullptr, stream);
}

extern "C" int wsr_conv3d_wgrad_parts_x2(const wsr_conv_t* c, const void* x, const void* x2, int32_t x2_ctot, int32_t x2_c0,
                                         const void* dy, float* parts, int64_t part_stride, int32_t n_parts, void* stream) {
  if (!conv_geom_ok_split(c, x2_c0) || !x || !x2 || !dy || !parts || n_parts <= 0 || x2_ctot <= 0) return WSR_EINVAL;
  if (part_stride < (int64_t)c->Cout * c->KX * c->KY * c->KZ * c->Cin) return WSR_EINVAL;
  return wgrad_any(c, x, dy, parts, 0, 0, (long)part_stride, n_parts, nullptr, stream, x2, x2_ctot, x2_c0);
}
