// Implicit-GEMM 3-D convolution on the MFMA 16x16 family (gfx950).
//
//   y[m, n] = sum_{tap, c} x[src(m, tap), c] * w[n, tap, c]
//
// m = output voxel (b, x, y, z) flattened, n = output channel.  K = (tap, c) is
// walked as a flat sequence of 16-byte "pieces" (8 bf16 / 4 fp32 channels), so a
// K-step may straddle taps and no K padding is wasted for Cin = 144 or 16.
// Operands are staged through LDS ([row][128 B + 16 B pad], conflict-free for
// ds_read_b128) with the next K-step's global loads in flight during the MFMAs.
// The MFMA is issued as D = W * X^T so that each lane ends up with 4 consecutive
// output channels of one voxel (vector stores into the NDHWC window).
//
// The same kernel serves the forward pass and the input-gradient pass: dgrad is
// the gather src = (dst + p - k) / s over the [Cin][taps][Cout] re-packed filter.
#include "common.h"

namespace {

constexpr int RS = 144;     // LDS row stride in bytes (128 data + 16 pad)
constexpr int PPS = 8;      // pieces per row per K-step (128 B)
constexpr int MAXTAPS = 128;

struct IgemmArgs {
  const char* in;
  const char* w;
  char* out;
  const float* bias;
  const float* chan_scale;
  const char* res;
  int res_ctot, res_off;
  float alpha, beta, slope;
  int act, out_planar;
  int B, Xi, Yi, Zi, Xo, Yo, Zo;
  int Cin, in_ctot, in_off;
  int Cout, out_ctot, out_off;
  int KX, KY, KZ;
  int mx, my, mz;  // row coordinate multiplier (stride for fwd, 1 for dgrad)
  int ox, oy, oz;  // tap offset base: fwd d = k - p ; dgrad d = p - k
  int tap_sign;    // +1 fwd, -1 dgrad
  int dx_, dy_, dz_;  // divisors (dgrad of strided conv), 1 otherwise
  int ups;         // nearest x(2,2,1) read
  int M;           // B*Xo*Yo*Zo
  int ppt;         // pieces per tap = Cin*sizeof(T)/16
  int total_pieces;
  int vec_ok;      // 4-channel vector epilogue allowed
  int pcls;        // strided dgrad: rows grouped by parity class (x%sx, y%sy, z%sz) so that a block's taps
  int tpc;         //   that divide are the same for all its rows and the others are skipped; tiles per class
};

template <class T, int WM, int WN, int TM, int TN, bool GENERAL>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmArgs a) {
  using elem = typename T::elem;
  constexpr int BM = WM * TM * 16;  // voxel rows per block
  constexpr int BN = WN * TN * 16;  // channel rows per block
  constexpr int XI = BM / 32;       // X rows staged per thread
  constexpr int WI = (BN + 31) / 32;
  static_assert(WM * WN == 4, "4 waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int4* rowc = reinterpret_cast<int4*>(smem);                      // [BM]
  int* taptab = reinterpret_cast<int*>(smem + BM * 16);            // [MAXTAPS] packed offsets of the taps walked
  int* tapid = taptab + MAXTAPS;                                   // [MAXTAPS] their index in the filter
  int* rowm = tapid + MAXTAPS;                                     // [BM] output voxel of a row (-1: none)
  int* ntap_s = rowm + BM;                                         // [4]
  char* Xt = smem + BM * 16 + MAXTAPS * 8 + BM * 4 + 16;           // [BM][RS]
  char* Wt = Xt + BM * RS;                                         // [BN][RS]

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int ntiles = (a.Cout + BN - 1) / BN;
  const int nwg = gridDim.x;
  const int bid = xcd_remap(blockIdx.x, nwg);
  const int mt = bid / ntiles, nt = bid % ntiles;
  const int n0 = nt * BN;
  const int taps = a.KX * a.KY * a.KZ;

  // parity class of this block (strided dgrad) and its voxel sub-grid
  int cx = 0, cy = 0, cz = 0, Xc = a.Xo, Yc = a.Yo, Zc = a.Zo, m0 = mt * BM;
  const bool pc = GENERAL && a.pcls;
  if (pc) {
    const int cls = mt / a.tpc;
    m0 = (mt - cls * a.tpc) * BM;
    cz = cls % a.dz_;
    cy = (cls / a.dz_) % a.dy_;
    cx = cls / (a.dz_ * a.dy_);
    Xc = (a.Xo - cx + a.dx_ - 1) / a.dx_;
    Yc = (a.Yo - cy + a.dy_ - 1) / a.dy_;
    Zc = (a.Zo - cz + a.dz_ - 1) / a.dz_;
  }
  const int Mc = a.B * Xc * Yc * Zc;

  // ---- per-block tables ------------------------------------------------------
  for (int r = t; r < BM; r += 256) {
    int m = m0 + r;
    int4 rc;
    int mo = -1;
    if (m < Mc) {
      int zo = m % Zc;
      int q = m / Zc;
      int yo = q % Yc;
      q /= Yc;
      int xo = q % Xc;
      int b = q / Xc;
      if (pc) {
        xo = xo * a.dx_ + cx;
        yo = yo * a.dy_ + cy;
        zo = zo * a.dz_ + cz;
      }
      rc.x = b * a.Xi * a.Yi * a.Zi;
      rc.y = xo * a.mx + a.ox;
      rc.z = yo * a.my + a.oy;
      rc.w = zo * a.mz + a.oz;
      mo = ((b * a.Xo + xo) * a.Yo + yo) * a.Zo + zo;
    } else {
      rc.x = 0;
      rc.y = -(1 << 28);  // fails every bounds test
      rc.z = 0;
      rc.w = 0;
    }
    rowc[r] = rc;
    rowm[r] = mo;
  }
  if (!pc) {
    for (int k = t; k < taps; k += 256) {
      int kz = k % a.KZ;
      int q = k / a.KZ;
      int ky = q % a.KY;
      int kx = q / a.KY;
      taptab[k] = ((a.tap_sign * kx) & 0x3ff) | (((a.tap_sign * ky) & 0x3ff) << 10) | (((a.tap_sign * kz) & 0x3ff) << 20);
      tapid[k] = k;
    }
    if (t == 0) ntap_s[0] = taps;
  } else if (t == 0) {  // only the taps whose gather coordinate divides by the stride for this class
    int n = 0;
    for (int k = 0; k < taps; ++k) {
      int kz = k % a.KZ;
      int q = k / a.KZ;
      int ky = q % a.KY;
      int kx = q / a.KY;
      // x = xo + ox + tap_sign*kx with xo == cx (mod sx)
      int rx = (cx + a.ox + a.tap_sign * kx) % a.dx_, ry = (cy + a.oy + a.tap_sign * ky) % a.dy_,
          rz = (cz + a.oz + a.tap_sign * kz) % a.dz_;
      if (rx == 0 && ry == 0 && rz == 0) {
        taptab[n] = ((a.tap_sign * kx) & 0x3ff) | (((a.tap_sign * ky) & 0x3ff) << 10) | (((a.tap_sign * kz) & 0x3ff) << 20);
        tapid[n] = k;
        ++n;
      }
    }
    ntap_s[0] = n;
  }
  __syncthreads();
  const int ntap = ntap_s[0];
  const int npieces = a.ppt * ntap;  // pieces of the K walk of this block

  // ---- staging state -----------------------------------------------------------
  const int q = t & (PPS - 1);  // piece slot inside the K-step
  const int rg = t >> 3;        // 0..31
  int tap = 0, c8 = q;          // (tap, piece-in-tap) of this thread's slot
  while (c8 >= a.ppt) {
    c8 -= a.ppt;
    ++tap;
  }
  int P = q;  // flat piece index of the walk
  const int nsteps = (npieces + PPS - 1) / PPS;

  uint4 xr[XI], wr[WI];

  auto load_step = [&]() {
    const bool pv = P < npieces;
    int dx = 0, dy = 0, dz = 0;
    if (pv) {
      int tt = taptab[tap];
      dx = (tt << 22) >> 22;
      dy = (tt << 12) >> 22;
      dz = (tt << 2) >> 22;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      int4 rc = rowc[rg + 32 * i];
      int x = rc.y + dx, y = rc.z + dy, z = rc.w + dz;
      bool ok = pv;
      if (GENERAL) {
        if (a.dx_ > 1) { ok = ok && (x % a.dx_ == 0); x /= a.dx_; }
        if (a.dy_ > 1) { ok = ok && (y % a.dy_ == 0); y /= a.dy_; }
        if (a.dz_ > 1) { ok = ok && (z % a.dz_ == 0); z /= a.dz_; }
        if (a.ups) {
          ok = ok && ((unsigned)x < (unsigned)(2 * a.Xi)) && ((unsigned)y < (unsigned)(2 * a.Yi));
          x >>= 1;
          y >>= 1;
        }
      }
      ok = ok && ((unsigned)x < (unsigned)a.Xi) && ((unsigned)y < (unsigned)a.Yi) && ((unsigned)z < (unsigned)a.Zi);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) {
        long vox = (long)rc.x + ((long)x * a.Yi + y) * a.Zi + z;
        const char* p = a.in + (vox * a.in_ctot + a.in_off) * (long)sizeof(elem) + (long)c8 * 16;
        v = *reinterpret_cast<const uint4*>(p);
      }
      xr[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      int r = rg + 32 * i;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r < BN && pv && (n0 + r) < a.Cout) {
        const char* p = a.w + ((long)(n0 + r) * a.total_pieces + (long)tapid[tap] * a.ppt + c8) * 16;
        v = *reinterpret_cast<const uint4*>(p);
      }
      wr[i] = v;
    }
    // advance to the next K-step
    P += PPS;
    c8 += PPS;
    while (c8 >= a.ppt) {
      c8 -= a.ppt;
      ++tap;
    }
  };

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  load_step();
  const int fr = lane & 15, fg = lane >> 4;
  for (int s = 0; s < nsteps; ++s) {
    __syncthreads();  // previous step's fragment reads are done
#pragma unroll
    for (int i = 0; i < XI; ++i) *reinterpret_cast<uint4*>(Xt + (rg + 32 * i) * RS + q * 16) = xr[i];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      int r = rg + 32 * i;
      if (r < BN) *reinterpret_cast<uint4*>(Wt + r * RS + q * 16) = wr[i];
    }
    __syncthreads();
    if (s + 1 < nsteps) load_step();  // in flight during the MFMAs
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 wf[TN], xf[TM];
#pragma unroll
      for (int j = 0; j < TN; ++j)
        wf[j] = *reinterpret_cast<const uint4*>(Wt + ((wn * TN + j) * 16 + fr) * RS + kk * 64 + fg * 16);
#pragma unroll
      for (int i = 0; i < TM; ++i)
        xf[i] = *reinterpret_cast<const uint4*>(Xt + ((wm * TM + i) * 16 + fr) * RS + kk * 64 + fg * 16);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_chunk<T>(acc[i][j], wf[j], xf[i]);
    }
  }

  // ---- epilogue -------------------------------------------------------------------
  // acc[i][j][reg]: channel = n0 + (wn*TN+j)*16 + fg*4 + reg ; voxel = m0 + (wm*TM+i)*16 + fr
  const long vox_per_b = (long)a.Xo * a.Yo * a.Zo;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = rowm[(wm * TM + i) * 16 + fr];
    if (m < 0) continue;
    const int b = (int)(m / vox_per_b);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int co0 = n0 + (wn * TN + j) * 16 + fg * 4;
      if (co0 >= a.Cout) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      const int nval = (a.Cout - co0) < 4 ? (a.Cout - co0) : 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r < nval) {
          float x = v[r];
          if (a.bias) x += a.bias[co0 + r];
          if (a.act) x = x > 0.f ? x : x * a.slope;
          if (a.chan_scale) x *= a.chan_scale[(long)b * a.Cout + co0 + r];
          v[r] = x * a.alpha;
        }
      }
      if (a.out_planar) {
        float* o = reinterpret_cast<float*>(a.out);
        const long vi = m - (long)b * vox_per_b;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < nval) o[((long)b * a.Cout + co0 + r) * vox_per_b + vi] = v[r];
      } else {
        elem* o = reinterpret_cast<elem*>(a.out) + (long)m * a.out_ctot + a.out_off + co0;
        const elem* rp = a.res ? reinterpret_cast<const elem*>(a.res) + (long)m * a.res_ctot + a.res_off + co0 : nullptr;
        if (nval == 4 && a.vec_ok) {
          float4 o4 = make_float4(v[0], v[1], v[2], v[3]);
          if (rp) {
            float4 r4 = ld4<T>(rp);
            o4.x += a.beta * r4.x;
            o4.y += a.beta * r4.y;
            o4.z += a.beta * r4.z;
            o4.w += a.beta * r4.w;
          }
          st4<T>(o, o4);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (r < nval) {
              float x = v[r];
              if (rp) x += a.beta * ldf<T>(rp + r);
              stf<T>(o + r, x);
            }
        }
      }
    }
  }
}

template <class T, int WM, int WN, int TM, int TN>
int launch_cfg(const IgemmArgs& a, bool general, hipStream_t st) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  int mtiles = (a.M + BM - 1) / BM;
  const int ntiles = (a.Cout + BN - 1) / BN;
  IgemmArgs b = a;
  if (general && a.pcls) {  // one set of row tiles per parity class
    const long mc = (long)a.B * ((a.Xo + a.dx_ - 1) / a.dx_) * ((a.Yo + a.dy_ - 1) / a.dy_) *
                    ((a.Zo + a.dz_ - 1) / a.dz_);
    b.tpc = (int)((mc + BM - 1) / BM);
    mtiles = b.tpc * a.dx_ * a.dy_ * a.dz_;
  }
  const size_t lds = BM * 16 + MAXTAPS * 8 + BM * 4 + 16 + (size_t)(BM + BN) * RS;
  dim3 grid(mtiles * ntiles), block(256);
  if (general)
    hipLaunchKernelGGL((igemm_kernel<T, WM, WN, TM, TN, true>), grid, block, lds, st, b);
  else
    hipLaunchKernelGGL((igemm_kernel<T, WM, WN, TM, TN, false>), grid, block, lds, st, b);
  WSR_LAUNCH_CHECK();
  return 0;
}

template <class T>
int launch_igemm(const IgemmArgs& a, bool general, hipStream_t st) {
  const int N = a.Cout;
  if (N <= 16) return launch_cfg<T, 4, 1, 4, 1>(a, general, st);   // 256 x 16
  if (N <= 32) return launch_cfg<T, 4, 1, 4, 2>(a, general, st);   // 256 x 32
  if (N <= 64) return launch_cfg<T, 4, 1, 2, 4>(a, general, st);   // 128 x 64
  if (N == 144) return launch_cfg<T, 4, 1, 2, 9>(a, general, st);  // 128 x 144
  return launch_cfg<T, 2, 2, 4, 4>(a, general, st);                // 128 x 128
}

int run_igemm(IgemmArgs& a, int dtype, bool general, hipStream_t st) {
  const int esz = dtype == WSR_BF16 ? 2 : 4;
  const int epp = 16 / esz;
  if (a.Cin % epp != 0 || a.in_ctot % epp != 0 || a.in_off % epp != 0) return WSR_EUNSUPPORTED;
  a.ppt = a.Cin / epp;
  a.total_pieces = a.ppt * a.KX * a.KY * a.KZ;
  a.vec_ok = (!a.out_planar && a.out_ctot % 4 == 0 && a.out_off % 4 == 0 &&
              (!a.res || (a.res_ctot % 4 == 0 && a.res_off % 4 == 0)))
                 ? 1
                 : 0;
  if ((long)a.B * a.Xo * a.Yo * a.Zo > 0x7fffffffL) return WSR_EUNSUPPORTED;
  return dtype == WSR_BF16 ? launch_igemm<BF16>(a, general, st) : launch_igemm<F32>(a, general, st);
}

}  // namespace

extern "C" int wsr_conv3d_fwd(const wsr_conv_t* c, const void* x, const void* w, void* y,
                              const wsr_epilogue_t* ep, void* stream) {
  if (!conv_geom_ok(c) || !x || !w || !y) return WSR_EINVAL;
  if (c->lat) return WSR_EUNSUPPORTED;  // parity convs of the sub-pixel form: tile kernels only
  IgemmArgs a{};
  a.in = (const char*)x;
  a.w = (const char*)w;
  a.out = (char*)y;
  a.alpha = 1.f;
  if (ep) {
    a.bias = ep->bias;
    a.chan_scale = ep->chan_scale;
    a.res = (const char*)ep->res;
    a.res_ctot = ep->res_ctot;
    a.res_off = ep->res_off;
    a.alpha = ep->alpha;
    a.beta = ep->beta;
    a.slope = ep->slope;
    if (ep->act > 1 || ep->act_c1 > 0 || ep->res2) return WSR_EUNSUPPORTED;  // tile / streaming kernels only
    a.act = ep->act;
    a.out_planar = ep->out_planar;
    if (a.res && (a.res_off < 0 || a.res_off + c->Cout > a.res_ctot)) return WSR_EINVAL;
  }
  a.B = c->B; a.Xi = c->Xi; a.Yi = c->Yi; a.Zi = c->Zi;
  a.Xo = c->Xo; a.Yo = c->Yo; a.Zo = c->Zo;
  a.Cin = c->Cin; a.in_ctot = c->in_ctot; a.in_off = c->in_off;
  a.Cout = c->Cout; a.out_ctot = c->out_ctot; a.out_off = c->out_off;
  a.KX = c->KX; a.KY = c->KY; a.KZ = c->KZ;
  a.mx = c->sx; a.my = c->sy; a.mz = c->sz;
  a.ox = -c->px; a.oy = -c->py; a.oz = -c->pz;
  a.tap_sign = 1;
  a.dx_ = a.dy_ = a.dz_ = 1;
  a.ups = c->upsample_xy ? 1 : 0;
  a.M = c->B * c->Xo * c->Yo * c->Zo;
  return run_igemm(a, c->dtype, a.ups != 0, as_stream(stream));
}

// dgrad: the "output" of the gather GEMM is dx (Cin channels at the conv's input
// resolution, or at the up-sampled resolution when upsample_xy), the reduction
// runs over (tap, Cout) of dy.
extern "C" int wsr_conv3d_dgrad(const wsr_conv_t* c, const void* dy, const void* wt, void* dx, float alpha,
                                int accumulate, int dx_planar, void* stream) {
  if (!conv_geom_ok(c) || !dy || !wt || !dx) return WSR_EINVAL;
  if (c->lat) return WSR_EUNSUPPORTED;  // parity convs of the sub-pixel form: tile kernels only
  const int ux = c->upsample_xy ? 2 : 1;
  IgemmArgs a{};
  a.in = (const char*)dy;
  a.w = (const char*)wt;
  a.out = (char*)dx;
  a.alpha = alpha;
  a.out_planar = dx_planar ? 1 : 0;
  if (accumulate) {
    if (dx_planar) return WSR_EUNSUPPORTED;
    a.res = (const char*)dx;
    a.res_ctot = c->in_ctot;
    a.res_off = c->in_off;
    a.beta = 1.f;
  }
  a.B = c->B;
  a.Xi = c->Xo; a.Yi = c->Yo; a.Zi = c->Zo;            // gathered tensor = dy
  a.Xo = c->Xi * ux; a.Yo = c->Yi * ux; a.Zo = c->Zi;  // produced tensor = dx
  a.Cin = c->Cout; a.in_ctot = c->out_ctot; a.in_off = c->out_off;
  a.Cout = c->Cin; a.out_ctot = c->in_ctot; a.out_off = c->in_off;
  a.KX = c->KX; a.KY = c->KY; a.KZ = c->KZ;
  a.mx = a.my = a.mz = 1;
  a.ox = c->px; a.oy = c->py; a.oz = c->pz;
  a.tap_sign = -1;
  a.dx_ = c->sx; a.dy_ = c->sy; a.dz_ = c->sz;
  a.ups = 0;
  a.M = c->B * a.Xo * a.Yo * a.Zo;
  const bool general = (c->sx | c->sy | c->sz) != 1;
  a.pcls = general ? 1 : 0;
  return run_igemm(a, c->dtype, general, as_stream(stream));
}
