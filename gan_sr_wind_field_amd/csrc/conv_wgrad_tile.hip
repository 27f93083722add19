// Filter gradient of a stride-1 3-D convolution from LDS-resident spatial tiles (bf16).
//
//   dw[n, tap, c] += sum_v dy[v, n] * x[v + tap - pad, c]
//
// A workgroup (8 waves) owns one (n-chunk of 16*TN output channels, c-chunk of
// 16*CT input channels) pair and walks a strided list of spatial tiles.  Per tile
//   Xs : the input tile WITH its halo, (TX+KX-1)(TY+KY-1)(TZ+KZ-1) voxels x 16*CT ch
//   Ys : the output-gradient tile, TX*TY*TZ voxels x 16*TN ch
// are brought into LDS once by LDS-DMA (global_load_lds), the NEXT tile's DMA in
// flight while the current one is contracted, and the contraction over the tile's
// voxels runs for ALL taps: the x operand of tap (kx,ky,kz) is the same LDS image
// read at a shifted voxel index, so x and dy leave L2/HBM once per tile, not once
// per tap.  Accumulators of every (tap, 16-channel c-tile) "slot" stay in
// registers across the whole tile list (slots are dealt round-robin to the waves:
// wavefront-level partial sums) and are added to the fp32 gradient once, at the end.
//
// LDS images are voxel-major rows so that the lanes of one DMA instruction fetch
// whole 32..128-byte runs of a voxel:  Xs = CT planes of 32-byte rows [c-tile][halo
// voxel][16 ch], Ys = 128-byte rows [voxel][8 octets] (the last 8-2*TN octets are
// padding).  The reduction index (voxel) is the slow dimension of both MFMA
// operands, so fragments come from the transposing read ds_read_b64_tr_b16.  The
// MFMA k index is mapped to voxels so that a half-wave reads 8 CONSECUTIVE voxels
// (k = 8G+j <-> voxel 4G + (j&3) + 16*(j>>2) of the 32-voxel step): for x that is
// 256 contiguous bytes at any tap shift - conflict free, and the shifted address is
// just base + tap offset; for dy the 32-byte blocks of a row are XOR-swizzled by
// (voxel>>1)&3 (on the DMA source side: the LDS destination of a DMA is linear).
//
// `tri_step` > 0 describes the block-triangular structure of a residual dense
// block: output channel n belongs to conv i = n / tri_step whose input is only
// channels [0, tri_base + i*tri_step) of the shared dense buffer, so the four
// growth convs of an RDB (reference torch_blocks.py:256-267) are ONE launch.
#include <cstdlib>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef s16x4_t __attribute__((address_space(3))) * lds_s16x4_ptr;

struct WgtArgs {
  const unsigned short* x;
  const unsigned short* x2;    // input channels >= x2_c0 come from channels [0, ...) of this tensor (x2_ctot per voxel): the
  int x2_ctot, x2_c0;          // generator's concat as two tensors (wsr_conv3d_wgrad_parts_x2); NULL = one tensor
  const unsigned short* dy;
  float* dw;
  const void* zero16;
  int B, Xi, Yi, Zi, Xo, Yo, Zo;
  int Cin, in_ctot, in_off;    // Cin = padded channel count of the x window (dw row length)
  int Cout, out_ctot, out_off; // Cout = channels of dy that are real (dw rows)
  int KX, KY, KZ, px, py, pz, ups;
  int TX, TY, TZ;              // output tile
  int n_chunks, c_chunks, S;   // grid = n_chunks * c_chunks * S
  int tiles_x, tiles_y, tiles_z, ntiles;
  int nbuf;                    // LDS tile buffers (2: next tile's DMA overlaps the MFMAs)
  int xp_bytes;                // one c-tile plane of the x image
  int xs_bytes, buf_bytes;     // Xs image size (CT planes), Xs + Ys size (per buffer)
  int off_buf;                 // LDS carve (bytes); htab sits at 0
  int tri_base, tri_step;
  int n_active;                 // > 0: only these (n-chunk, c-chunk) pairs have work (block-triangular launches)
  unsigned char act_nc[64], act_cc[64];
  long part_stride;             // > 0: spatial split s STORES its sums at dw + s*part_stride (no atomics, see
                                // wsr_conv3d_wgrad_parts); 0: every split adds into dw with float atomics
  int yl_m, yl_ox, yl_oy;       // dy is the sub-lattice (m*x + ox, m*y + oy, z) of a tensor m times as large along x and y
                                // (parity convs of a sub-pixel up-sampling conv, wsr_conv_t.lat); m = 1: dy itself
  int xl_m, xl_ox, xl_oy;       // x is the sub-lattice (m*x + ox, m*y + oy, mz*z + oz) of a tensor m (mz) times as large
  int xl_mz, xl_oz;             // (filter gradients of the stride-2 down-sampling convs in parity form, wsr_conv_t.lat = 3)
  int prio;                     // 1: waves 4..7 at s_setprio 1 in the tile loop (tuning switch)
  int S_forced;                 // > 0: the number of spatial splits the caller was told (wsr_conv3d_wgrad_nparts)
  int plan_only;                // host side: compute the launch geometry (S) and return without launching
  unsigned long long* stamps;  // -DWSR_CT_STAMPS builds: [workgroup][8] clock samples (else unused)
  int ablate;                  // -DWSR_CT_STAMPS builds, timing only: skip 1 = tile DMA, 4 = LDS reads, 8 = MFMAs
};

__device__ __forceinline__ uint4 tr_frag(const char* lo, const char* hi) {
  s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lo));
  s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(hi));
  uint2 l2 = __builtin_bit_cast(uint2, a), h2 = __builtin_bit_cast(uint2, b);
  return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// LDS-DMA of 16 B per lane: LDS[lds_addr + 16*lane] <- *gsrc (see conv_tile.hip: inline asm keeps hipcc
// from draining it in front of every LDS read; waited for explicitly with dma_wait()).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_addr)
      : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most n (uniform, run-time) of this wave's DMA units are still in flight
__device__ __forceinline__ void dma_wait_upto(int n) {
#define WG_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    WG_VM(0) WG_VM(1) WG_VM(2) WG_VM(3) WG_VM(4) WG_VM(5) WG_VM(6) WG_VM(7) WG_VM(8) WG_VM(9) WG_VM(10) WG_VM(11)
    WG_VM(12) WG_VM(13) WG_VM(14) WG_VM(15) WG_VM(16) WG_VM(17) WG_VM(18) WG_VM(19) WG_VM(20) WG_VM(21) WG_VM(22)
    WG_VM(23) WG_VM(24) WG_VM(25) WG_VM(26) WG_VM(27) WG_VM(28) WG_VM(29) WG_VM(30) WG_VM(31) WG_VM(32) WG_VM(33)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#undef WG_VM
}

// 32-byte block swizzle of the dy image: 8 consecutive voxels x one n-tile -> 8 distinct 32-byte bank slots
// (128-byte rows, TN <= 4: two voxels per 256-byte bank row; 256-byte rows, TN = 8: one)
#ifdef WSR_CT_STAMPS
#define WG_STAMP(k)                                                                                    \
  do {                                                                                                 \
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + (k)] = (k) >= 6 ? clock64() : wall_clock64(); \
  } while (0)
#else
#define WG_STAMP(k) do {} while (0)
#endif

// (TN = 1: 32-byte rows like the x image - one block per row, 8 consecutive voxels are 256 contiguous bytes: no swizzle)
template <int TN> __device__ __forceinline__ int ysw(int v) { return TN == 1 ? 0 : (TN <= 4 ? (v >> 1) & 3 : v & 7); }
// bytes per dy row in LDS.  One n-tile (the thin convs: the z-folded last conv, the terrain convs) used to take the
// 128-byte rows of the 2..4-tile form, three quarters of them zeros fetched from the zero page: 32 KB of a 68 KB tile
// DMA on the last conv's gradient, which is bound by exactly that stream.
constexpr int wgt_rby(int tn) { return tn == 1 ? 32 : (tn <= 4 ? 128 : 256); }

// DMA units (1 KB) a wave may have to issue per tile for the x / dy image: registers of the resolved source geometry.
// The accumulator-heavy instantiations (5x5x5, 192 accumulators) get what their 4x4x16 tile needs and no more.
constexpr int wgt_xk(int spw, int tn) { return spw * tn * 4 >= 192 ? 5 : 6; }
constexpr int wgt_yk(int spw, int tn) { return spw * tn * 4 >= 192 ? 4 : 5; }

template <int TN, int SPW, int CT, bool Z16>
__global__ __launch_bounds__(512) void wgrad_tile_kernel(const WgtArgs a) {
  constexpr int WAVES = 8, NT = 512;
  constexpr int XRPU = 32;            // x rows (32 B) per 1 KB DMA unit
  constexpr int XK = wgt_xk(SPW, TN), YK = wgt_yk(SPW, TN);  // DMA units per wave per tile (checked on the host)
  constexpr int RBY = wgt_rby(TN);          // bytes per dy row
  constexpr int YRPU = 1024 / RBY;          // dy rows per 1 KB DMA unit
  static_assert(TN <= 8, "dy rows hold at most 128 channels");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  WG_STAMP(0);
  WG_STAMP(6);

  const int Ly = a.TY + a.KY - 1, Lz = a.TZ + a.KZ - 1, Lx = a.TX + a.KX - 1;
  const int L = Lx * Ly * Lz;
  const int M = a.TX * a.TY * a.TZ;
  const int taps = a.KX * a.KY * a.KZ;
  unsigned short* htab = reinterpret_cast<unsigned short*>(smem);  // [M] halo index of voxel m
  char* buf0 = smem + a.off_buf;
  typedef __attribute__((address_space(3))) char* lptr_t;
  const unsigned buf_lds = (unsigned)(unsigned long)(lptr_t)buf0;

  // ---- which chunk pair / spatial slice ------------------------------------------
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  int cc, nc, s0;
  if (a.n_active > 0) {  // pairs the triangular structure leaves empty are not launched at all
    const int pr = bid % a.n_active;
    s0 = bid / a.n_active;
    cc = a.act_cc[pr];
    nc = a.act_nc[pr];
  } else {
    cc = bid % a.c_chunks;
    bid /= a.c_chunks;
    nc = bid % a.n_chunks;
    s0 = bid / a.n_chunks;
  }
  const int c0 = cc * 16 * CT, n0 = nc * 16 * TN;
  // the tensor this workgroup's c-chunk lives in (uniform; the host made x2_c0 a multiple of the chunk)
  const bool second = a.x2 != nullptr && c0 >= a.x2_c0;
  const unsigned short* xt = second ? a.x2 : a.x;
  const int x_ctot = second ? a.x2_ctot : a.in_ctot;
  const int x_off = second ? c0 - a.x2_c0 : a.in_off + c0;

  // block-triangular structure: n-tile i is needed iff c0 < tri_base + tri_step*conv(n)
  bool act[TN];
  bool any = false;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + 16 * i;
    bool ok = n < a.Cout;
    if (ok && a.tri_step > 0) {  // the widest-input conv among the tile's channels decides
      const int n_last = (n + 15 < a.Cout) ? n + 15 : a.Cout - 1;
      ok = c0 < a.tri_base + a.tri_step * (n_last / a.tri_step);
    }
    act[i] = ok;
    any = any || ok;
  }
  if (!any) return;  // uniform over the workgroup

  for (int m = t; m < M; m += NT) {
    const int oz = m % a.TZ, r = m / a.TZ;
    const int oy = r % a.TY, ox = r / a.TY;
    htab[m] = (unsigned short)((ox * Ly + oy) * Lz + oz);
  }

  // ---- DMA geometry of this lane, resolved once (tiles differ only in their origin) ------------
  // Per 1 KB unit k: geo = x | y<<8 | z<<16 | valid<<24 of the (halo) voxel this lane fetches, and rel = its
  // element offset from the tile's first (halo) voxel, channel included.  Per tile only a scalar base, the
  // add, and - for tiles that touch the volume border - three range checks remain.
  const int U = a.ups ? 1 : 0;
  const int parx = U ? ((-a.px) & 1) : 0, pary = U ? ((-a.py) & 1) : 0;  // parity of x0 - px (TX, TY even when up-sampled)
  const int XUP = (L + XRPU - 1) / XRPU;  // 1 KB units per c-tile plane of the x image
  const int XU = XUP * CT;
  const int YU = (M + YRPU - 1) / YRPU;   // ... of the dy image
  unsigned xgeo[XK], ygeo[YK];
  int xrel[XK], yrel[YK];
#pragma unroll
  for (int k = 0; k < XK; ++k) {
    const int u = wave + WAVES * k;
    unsigned geo = 0;
    int rel = 0;
    if (u < XU) {
      const int ct = u / XUP;
      const int h = (u - ct * XUP) * XRPU + (lane >> 1);  // halo voxel = LDS row of plane ct
      const int ch8 = 2 * ct + (lane & 1);
      if (h < L && c0 + 8 * ch8 < a.Cin) {
        const int hz = h % Lz, qq = h / Lz;
        const int hy = qq % Ly, hx = qq / Ly;
        geo = hx | (hy << 8) | (hz << 16) | (1u << 24);
        rel = ((((hx + parx) >> U) * a.xl_m * (a.Yi * a.xl_m) + ((hy + pary) >> U) * a.xl_m) * (a.Zi * a.xl_mz) +
               hz * a.xl_mz) * x_ctot + 8 * ch8;
      }
    }
    xgeo[k] = geo;
    xrel[k] = rel;
  }
#pragma unroll
  for (int k = 0; k < YK; ++k) {
    const int u = wave + WAVES * k;
    unsigned geo = 0;
    int rel = 0;
    if (u < YU) {
      const int v = u * YRPU + lane / (RBY / 16), sl = lane % (RBY / 16);
      const int b32 = (sl >> 1) ^ ysw<TN>(v);
      const int oct = 2 * b32 + (sl & 1);
      // channel windows are whole octets here (the host routes anything else to the per-tap kernel)
      if (v < M && b32 < TN && n0 + 8 * oct < a.Cout) {
        const int oz = v % a.TZ, qq = v / a.TZ;
        const int oy = qq % a.TY, ox = qq / a.TY;
        geo = ox | (oy << 8) | (oz << 16) | (1u << 24);
        rel = ((ox * a.yl_m * (a.Yo * a.yl_m) + oy * a.yl_m) * a.Zo + oz) * a.out_ctot + 8 * oct;
      }
    }
    ygeo[k] = geo;
    yrel[k] = rel;
  }

  const unsigned short* zsrc = reinterpret_cast<const unsigned short*>(a.zero16);
  auto issue_tile = [&](int tile, int buf) {
    int r = tile;
    const int tz = r % a.tiles_z; r /= a.tiles_z;
    const int ty = r % a.tiles_y; r /= a.tiles_y;
    const int tx = r % a.tiles_x;
    const int b = r / a.tiles_x;
    const int x0 = tx * a.TX, y0 = ty * a.TY, z0 = tz * a.TZ;
    const unsigned dstx = buf_lds + buf * a.buf_bytes;
    const unsigned dsty = dstx + a.xs_bytes;
    {  // x image with halo: halo voxel h sits at (x0 - px + hx, ...) of the (up-sampled) input
      const int lox = a.px - x0, loy = a.py - y0, loz = a.pz - z0;  // first in-range halo coordinate
      const int sx_ = a.Xi << U, sy_ = a.Yi << U;
      const bool inner = lox <= 0 && Lx - lox <= sx_ && loy <= 0 && Ly - loy <= sy_ && loz <= 0 && Lz - loz <= a.Zi;
      // (xl_m = xl_mz = 1 unless the input sits on a lattice; then U = 0)
      const long base = ((((long)b * a.Xi * a.xl_m + (long)((x0 - a.px - parx) >> U) * a.xl_m + a.xl_ox) * (a.Yi * a.xl_m) +
                          (long)((y0 - a.py - pary) >> U) * a.xl_m + a.xl_oy) * (a.Zi * a.xl_mz) +
                         (long)(z0 - a.pz) * a.xl_mz + a.xl_oz) * x_ctot + x_off;
      const unsigned short* bp = xt + base;
#pragma unroll
      for (int k = 0; k < XK; ++k) {
        const int u = wave + WAVES * k;
#ifdef WSR_CT_STAMPS
        // (tuning build, ablate & 64 / & 128: every second / three of four x-image units are not fetched - the timing of a
        // launch whose halo image costs half / a quarter of the DMA, e.g. x-planes shared between neighbouring tiles)
        if ((a.ablate & 64) && (k & 1)) continue;
        if ((a.ablate & 128) && (k & 3)) continue;
#endif
        if (u < XU) {
          const unsigned geo = xgeo[k];
          bool ok = (geo >> 24) & 1;
          if (!inner)
            ok = ok && (unsigned)((int)(geo & 255) - lox) < (unsigned)sx_ &&
                 (unsigned)((int)((geo >> 8) & 255) - loy) < (unsigned)sy_ &&
                 (unsigned)((int)((geo >> 16) & 255) - loz) < (unsigned)a.Zi;
          glds16(ok ? bp + xrel[k] : zsrc, __builtin_amdgcn_readfirstlane(dstx + u * 1024));
        }
      }
    }
    {  // dy image: tile voxels only
      const int hix = a.Xo - x0, hiy = a.Yo - y0, hiz = a.Zo - z0;
      const bool inner = a.TX <= hix && a.TY <= hiy && a.TZ <= hiz;
      const long base = ((((long)b * a.Xo * a.yl_m + x0 * a.yl_m + a.yl_ox) * (a.Yo * a.yl_m) + y0 * a.yl_m + a.yl_oy) *
                             a.Zo + z0) * a.out_ctot + a.out_off + n0;
      const unsigned short* bp = a.dy + base;
#pragma unroll
      for (int k = 0; k < YK; ++k) {
        const int u = wave + WAVES * k;
        if (u < YU) {
          const unsigned geo = ygeo[k];
          bool ok = (geo >> 24) & 1;
          if (!inner)
            ok = ok && (int)(geo & 255) < hix && (int)((geo >> 8) & 255) < hiy && (int)((geo >> 16) & 255) < hiz;
          glds16(ok ? bp + yrel[k] : zsrc, __builtin_amdgcn_readfirstlane(dsty + u * 1024));
        }
      }
    }
  };

  // ---- this wave's slots: (tap, c-tile) pairs ------------------------------------------
  const int nslots = taps * CT;
  int soff[SPW];  // byte offset of the slot's x rows relative to the un-shifted tap of c-tile 0
#pragma unroll
  for (int j = 0; j < SPW; ++j) {
    const int sj = wave + WAVES * j;
    int off = 0, ct = 0;
    if (sj < nslots) {
      const int tap = sj / CT;
      ct = sj % CT;
      const int kz = tap % a.KZ, r = tap / a.KZ;
      const int ky = r % a.KY, kx = r / a.KY;
      off = (kx * Ly + ky) * Lz + kz;
    }
    soff[j] = __builtin_amdgcn_readfirstlane(off * 32 + ct * a.xp_bytes);
  }

  f32x4_t acc[SPW][TN];
#pragma unroll
  for (int j = 0; j < SPW; ++j)
#pragma unroll
    for (int i = 0; i < TN; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // tr-read lane roles: 16-lane group G supplies rows (voxels) 4G+q (+16), columns 4p..4p+3 of a 16-wide tile
  const int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int ksteps = M >> 5;

  WG_STAMP(1);
#ifdef WSR_CT_STAMPS
  long long st_bar = 0;
  int st_tiles = 0;
#endif
  // Operand pipeline of the contraction.  One slot is only TN MFMAs (64..128 matrix-pipe cycles) and an LDS
  // round trip is longer than that, so nothing may be requested right before it is used: the x fragment of
  // slot j+D and, one per slot, the dy fragments of the NEXT K-step are requested before the MFMAs of slot j
  // (ring of R registers sets, R | SPW so that every K-step starts at ring slot 0; the scheduler is fenced
  // because it otherwise sinks the requests back to their uses).  The last K-step of a tile prefetches a
  // dummy (itself) instead of branching: uniform branches would cost the counted waits.
  // (accumulator-heavy instantiations - the 5x5x5 kernels with 3 n-tiles - have no registers left for a
  // second dy set and a ring: they fetch per slot, LEAN)
  constexpr bool LEAN = SPW * TN * 4 >= 192;
  constexpr int R = SPW % 7 == 0 ? 7 : (SPW % 4 == 0 ? 4 : (SPW % 3 == 0 ? 3 : 1));
#ifdef WSR_WG_D
  constexpr int D = R > WSR_WG_D ? WSR_WG_D : R - 1;
#else
  constexpr int D = R > 3 ? 3 : R - 1;
#endif
  struct KP { const char *xlo, *xhi, *rlo, *rhi; int slo, shi; };
  // Voxel <-> MFMA k mapping of a 32-voxel K-step.  Z16 (tile z extent a multiple of 16, every shipped
  // shape): lane group G takes voxels 16*(G>>1) + 4*(G&1) + q and the same + 8 - both in ONE z column, so the
  // second transposing read of a fragment is the first one's address + a compile-time constant (8 rows:
  // 256 B in the x image, 8 dy rows with an unchanged swizzle) and costs no address arithmetic.  Otherwise
  // voxels 4G + q and + 16 through the halo table.
  auto kp_of = [&](const char* Xs, const char* Ys, int ks) {
    const int m_lo = Z16 ? ks * 32 + 16 * (G >> 1) + 4 * (G & 1) + q : ks * 32 + 4 * G + q;
    const int m_hi = m_lo + (Z16 ? 8 : 16);
    KP k;
    k.xlo = Xs + (int)htab[m_lo] * 32 + p * 8;
    k.rlo = Ys + m_lo * RBY + p * 8;
    k.slo = ysw<TN>(m_lo);
    if constexpr (Z16) {
      k.xhi = k.xlo + 8 * 32;
      k.rhi = k.rlo + 8 * RBY;
      k.shi = k.slo;
    } else {
      k.xhi = Xs + (int)htab[m_hi] * 32 + p * 8;
      k.rhi = Ys + m_hi * RBY + p * 8;
      k.shi = ysw<TN>(m_hi);
    }
    return k;
  };
  constexpr int XHI = 8 * 32, RHI = 8 * RBY;
  auto af_of = [&](const KP& k, int i) {
#ifdef WSR_CT_STAMPS
    if (a.ablate & 4) return make_uint4(lane, 1u, 2u, 3u);
#endif
    if constexpr (Z16) {
      const char* lo = k.rlo + ((i ^ k.slo) << 5);
      return tr_frag(lo, lo + RHI);
    } else {
      return tr_frag(k.rlo + ((i ^ k.slo) << 5), k.rhi + ((i ^ k.shi) << 5));
    }
  };
  auto bf_of = [&](const KP& k, int j) {
#ifdef WSR_CT_STAMPS
    if (a.ablate & 4) return make_uint4(lane, 1u, 2u, 3u);
#endif
    if constexpr (Z16) {
      const char* lo = k.xlo + soff[j];
      return tr_frag(lo, lo + XHI);
    } else {
      return tr_frag(k.xlo + soff[j], k.xhi + soff[j]);
    }
  };

  if (a.prio && wave >= WAVES / 2) __builtin_amdgcn_s_setprio(1);
  // Software pipeline over the tile list, a ring of a.nbuf buffers (DT = nbuf - 1 tiles ahead): iteration `it` requests
  // tile s0 + it*S into buffer it % nbuf while tile s0 + (it-DT)*S is contracted (one DMA call site, one MFMA call
  // site).  Two buffers in every shipped launch - a tile's MFMAs and transposing reads outlast its DMA; deeper rings
  // (counted waits: every tile is the same number of DMA units per wave) are a tuning switch, see launch_tile.
  const int DT = a.nbuf - 1;
  int upt = 0;  // DMA units this wave issues per tile
#pragma unroll
  for (int k = 0; k < XK; ++k) upt += (wave + WAVES * k < XU) ? 1 : 0;
#pragma unroll
  for (int k = 0; k < YK; ++k) upt += (wave + WAVES * k < YU) ? 1 : 0;
  const int ntl = (a.ntiles - s0 + a.S - 1) / a.S;  // tiles of this workgroup
  int bi = 0, bc = 0;                                // ring positions: next to fill, next to contract
  for (int it = 0;; ++it) {
    const int pre = s0 + it * a.S;
#ifdef WSR_CT_STAMPS
    if (pre < a.ntiles && !((a.ablate & 1) && it > 0)) issue_tile(pre, bi);
#else
    if (pre < a.ntiles) issue_tile(pre, bi);
#endif
    bi = bi + 1 == a.nbuf ? 0 : bi + 1;
    if (it >= DT) {
      const char* Xs = buf0 + bc * a.buf_bytes;
      const char* Ys = Xs + a.xs_bytes;
      bc = bc + 1 == a.nbuf ? 0 : bc + 1;
      if constexpr (LEAN) {
        // One x fragment ahead (ring of two): the transposing reads of slot j+1 are in flight during the MFMAs of
        // slot j; the scheduler is fenced, or it sinks the request back to its use and every slot pays an LDS round
        // trip (counters before: waves parked at s_waitcnt half of the time).  8.65 -> 8.0 ms on the 5x5x5 144 -> 144
        // gradient.  Measured and NOT kept: carrying the ring and the dy fragments across K-steps (in-place reload
        // after the last use: +3 %, a second dy set: spills) - the plain per-K-step form is the fastest; a ring of
        // three with the DMA geometry decoded again per border tile instead of held in 9 registers (+3 %: the
        // allocator fills the 256 registers either way and spills 24 bytes instead of 8).
        for (int ks = 0; ks < ksteps; ++ks) {
          const KP k = kp_of(Xs, Ys, ks);
          uint4 af[TN], bfr[2];
#pragma unroll
          for (int i = 0; i < TN; ++i) af[i] = af_of(k, i);
          bfr[0] = bf_of(k, 0);
#pragma unroll
          for (int j = 0; j < SPW; ++j) {
#ifdef WSR_CT_STAMPS
            // (tuning build, ablate & 16: only every fourth slot reads its x fragment, the others copy the previous
            // one - wrong sums, the timing of a kernel that derives the kz-shifted fragments in registers)
            if ((a.ablate & 16) && ((j + 1) & 3)) { if (j + 1 < SPW) bfr[(j + 1) & 1] = bfr[j & 1]; } else
#endif
            if (j + 1 < SPW) bfr[(j + 1) & 1] = bf_of(k, j + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TN; ++i)
              acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[i]),
                                                                  __builtin_bit_cast(bf16x8_t, bfr[j & 1]), acc[j][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
      KP kc = kp_of(Xs, Ys, 0);
      uint4 afA[TN], afB[TN], ring[R], nb;
#pragma unroll
      for (int i = 0; i < TN; ++i) afA[i] = af_of(kc, i);
#pragma unroll
      for (int d = 0; d < (R > 1 ? D : 1); ++d) ring[d] = bf_of(kc, d);
      // one K-step: consumes afc, requests afn (the next K-step's dy fragments)
      auto step = [&](int ks, uint4 (&afc)[TN], uint4 (&afn)[TN]) {
        const KP kn = kp_of(Xs, Ys, ks + 1 < ksteps ? ks + 1 : ks);
        if constexpr (R == 1) nb = bf_of(kn, 0);

#pragma unroll
        for (int j = 0; j < SPW; ++j) {
          if constexpr (R > 1) {
            const int f = j + D;
            if (f < SPW) ring[f % R] = bf_of(kc, f);
            else ring[f % R] = bf_of(kn, f - SPW);
          }
          if (j < TN) afn[j] = af_of(kn, j);
          __builtin_amdgcn_sched_barrier(0);
          // branch-free over slots and n-tiles: a slot past the end re-reads tap 0 into accumulators that are
          // never flushed, an n-tile the triangular structure does not need is computed and dropped at the flush
#ifdef WSR_CT_STAMPS
          if (a.ablate & 8) continue;
#endif
#pragma unroll
          for (int i = 0; i < TN; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, afc[i]),
                                                                __builtin_bit_cast(bf16x8_t, ring[j % R]), acc[j][i], 0, 0,
                                                                0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (SPW < TN) {  // (1x1x1 kernels: more n-tiles than slots)
#pragma unroll
          for (int i = SPW; i < TN; ++i) afn[i] = af_of(kn, i);
        }
        if constexpr (R == 1) ring[0] = nb;
        kc = kn;
      };
      for (int ks = 0; ks < ksteps; ks += 2) {
        step(ks, afA, afB);
        if (ks + 1 < ksteps) step(ks + 1, afB, afA);
      }
      }
    }
#ifdef WSR_CT_STAMPS
    const long long tw0 = clock64();
#endif
    {  // the tile contracted next (it - DT + 1) must have landed; the younger ones stay in flight
      const int tnext = it - DT + 1;
      if (tnext >= 0) {
        int later = ntl - 1 - tnext;
        later = later < 0 ? 0 : (later > it - tnext ? it - tnext : later);
        dma_wait_upto(later * upt);
      }
      __syncthreads();  // ... for everybody; the buffer just read is free again
#ifdef WSR_CT_STAMPS
      st_bar += clock64() - tw0;
      ++st_tiles;
#endif
      if (tnext >= 0 && tnext >= ntl) break;
    }
  }
  WG_STAMP(2);

  // ---- this workgroup's partial sums: acc[j][i][r] -> n = n0+16i+4G+r, c = c0+16ct+(lane&15); stored to the
  // split's own copy (deterministic two-pass form) or added to the shared one
  float* const dwp = a.dw + (long)s0 * a.part_stride;
  const bool store = a.part_stride > 0;
#pragma unroll
  for (int j = 0; j < SPW; ++j) {
    const int sj = wave + WAVES * j;
    if (sj >= nslots) continue;
    const int tap = sj / CT, ct = sj % CT;
    const int c = c0 + 16 * ct + (lane & 15);
    if (c >= a.Cin) continue;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      if (!act[i]) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 16 * i + 4 * G + r;
        if (n < a.Cout) {
          float* q = dwp + ((long)n * taps + tap) * a.Cin + c;
          if (store) *q = acc[j][i][r];
          else atomicAdd(q, acc[j][i][r]);
        }
      }
    }
  }
  WG_STAMP(3);
  WG_STAMP(7);
#ifdef WSR_CT_STAMPS
  if (a.stamps && threadIdx.x == 0) {
    a.stamps[(size_t)blockIdx.x * 8 + 4] = (unsigned long long)st_bar;
    a.stamps[(size_t)blockIdx.x * 8 + 5] = (unsigned long long)st_tiles;
  }
#endif
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

__device__ uint4 g_wg_zero16 = {0u, 0u, 0u, 0u};

template <int TN, int SPW, int CT>
int launch_tile(WgtArgs& a, hipStream_t st) {
  constexpr int WAVES = 8;
  const int taps = a.KX * a.KY * a.KZ;
  if (taps * CT > WAVES * SPW) return WSR_EUNSUPPORTED;
  if (a.x2 && a.x2_c0 % (16 * CT)) return WSR_EUNSUPPORTED;  // a workgroup's c-chunk lives in ONE of the two tensors
  {
    static void* zp = nullptr;
    if (!zp) {
      hipError_t e = hipGetSymbolAddress(&zp, HIP_SYMBOL(g_wg_zero16));
      if (e != hipSuccess) return (int)e;
    }
    a.zero16 = zp;
  }
  // tile: z whole when short, else 16 (the two z-octets of a half-wave read stay in one row run); x, y as
  // large as two LDS buffers and the per-wave DMA unit registers allow
  int tz = a.Zo <= 16 ? a.Zo : ((a.Zo % 16 == 0 || a.Zo > 64) ? 16 : 8);
  // no taps along z (the z-folded last conv, 5x5x1): a flat tile has no z halo and a far smaller x-y one
  // (8x8x4: 2.25x its voxels; 2x4x16: 6x) - that launch is bound by the halo re-reads of its 144-channel input
  if (a.KZ == 1 && a.KX * a.KY > 1 && tz > 4 && a.Zo % 4 == 0 && !WSR_ENV_SET("WSR_CT_NOFLAT")) tz = 4;
  static const int cand[][2] = {{8, 8}, {4, 8}, {4, 4}, {2, 4}, {2, 2}, {1, 2}, {1, 1}};
  int best = -1, best_nbuf = 0;
  // (two buffers.  WSR_WG_NBUF = 3..: a deeper ring of smaller tiles for the 1x1x1 gradient, see the kernel's pipeline
  // comment - measured SLOWER, 55 us against 46 at four buffers of 64 voxels.  Ablation stamps of that launch: tile
  // loop 34 us; without LDS reads and MFMAs still 27.6 us - the LDS-DMA stream itself, 134 MB at 5.3 TB/s, dy once per
  // c-chunk - and without the DMA 21.5 us: it is bound by DMA THROUGHPUT, not by the latency of one tile in flight)
  // (<3,4,1>, the exchanged thin gradient: 50 KB per buffer, a ring of three fits - WSR_WG_NBUF_THIN=3 measured equal,
  // 390 us either way: that launch is not waiting for one tile's DMA either)
  const int nbuf_want = taps == 1 ? WSR_ENV_INT("WSR_WG_NBUF", 2) : ((SPW == 4 && CT == 1) ? WSR_ENV_INT("WSR_WG_NBUF_THIN", 2) : 2);
  const int mmax = taps == 1 && nbuf_want > 2 ? WSR_ENV_INT("WSR_WG_MMAX", 64) : 1 << 30;
  for (int nbuf = nbuf_want; nbuf >= 2 && best < 0; --nbuf) {
    for (int ci = 0; ci < 7; ++ci) {
      const int tx = cand[ci][0], ty = cand[ci][1];
      const int M = tx * ty * tz;
      if ((M & 31) || M > mmax) continue;
      if (a.ups && ((tx | ty) & 1)) continue;  // the x0 - px parity must not depend on the tile
      const int L = (tx + a.KX - 1) * (ty + a.KY - 1) * (tz + a.KZ - 1);
      const int xs = CT * round_up(L * 32, 1024), ys = round_up(M * wgt_rby(TN), 1024);
      if (xs / 1024 > wgt_xk(SPW, TN) * WAVES || ys / 1024 > wgt_yk(SPW, TN) * WAVES || L > 65535) continue;
      if (round_up(M * 2, 1024) + nbuf * (xs + ys) > 160 * 1024) continue;
      best = ci;
      best_nbuf = nbuf;
      break;
    }
  }
  if (best < 0) return WSR_EUNSUPPORTED;
  a.TX = cand[best][0]; a.TY = cand[best][1]; a.TZ = tz;
  a.nbuf = best_nbuf;
  const int M = a.TX * a.TY * a.TZ;
  const int L = (a.TX + a.KX - 1) * (a.TY + a.KY - 1) * (a.TZ + a.KZ - 1);
  a.xp_bytes = round_up(L * 32, 1024);
  a.xs_bytes = CT * a.xp_bytes;
  a.buf_bytes = a.xs_bytes + round_up(M * wgt_rby(TN), 1024);
  a.off_buf = round_up(M * 2, 1024);
  const size_t lds = (size_t)a.off_buf + (size_t)a.nbuf * a.buf_bytes;
  a.n_chunks = (a.Cout + 16 * TN - 1) / (16 * TN);
  a.c_chunks = (a.Cin + 16 * CT - 1) / (16 * CT);
  a.tiles_x = (a.Xo + a.TX - 1) / a.TX;
  a.tiles_y = (a.Yo + a.TY - 1) / a.TY;
  a.tiles_z = (a.Zo + a.TZ - 1) / a.TZ;
  a.ntiles = a.B * a.tiles_x * a.tiles_y * a.tiles_z;
  int combos = a.n_chunks * a.c_chunks;
  a.n_active = 0;
  if (a.tri_step > 0 && combos <= 64) {  // same predicate as the kernel's `act`
    for (int nc = 0; nc < a.n_chunks; ++nc)
      for (int cc = 0; cc < a.c_chunks; ++cc) {
        bool any = false;
        for (int i = 0; i < TN; ++i) {
          const int n = nc * 16 * TN + 16 * i;
          if (n >= a.Cout) continue;
          const int n_last = (n + 15 < a.Cout) ? n + 15 : a.Cout - 1;
          any = any || cc * 16 * CT < a.tri_base + a.tri_step * (n_last / a.tri_step);
        }
        if (any) {
          a.act_nc[a.n_active] = (unsigned char)nc;
          a.act_cc[a.n_active] = (unsigned char)cc;
          ++a.n_active;
        }
      }
    if (a.n_active == 0) { a.S = 0; return 0; }
    combos = a.n_active;
  }
  // Spatial split S: ONE round of workgroups (one is resident per CU: 148 KB of LDS).  Measured on the
  // dense-block and the 5x5x5 wgrad: launches of <= 256 workgroups are fastest, a launch just over a
  // multiple of 256 is up to 1.6x slower (a nearly empty extra round), and more, smaller workgroups
  // only add accumulator flushes (each flush is worth ~6 tiles of MFMA work at the atomic rate).
  int S = 256 / combos;
  if (S < 1) S = 1;
  if (S > a.ntiles) S = a.ntiles;
  a.S = S;
  if (WSR_ENV_SET("WSR_WGRAD_S")) {  // tuning aid
    const int v = WSR_ENV_RAW("WSR_WGRAD_S");
    if (v >= 1 && v <= a.ntiles) a.S = v;
  }
  if (a.plan_only) return 0;
  if (a.part_stride > 0 && a.S_forced != a.S) return WSR_EINVAL;  // the caller sized `parts` for another split
  const bool z16 = a.TZ % 16 == 0;
  auto kern = z16 ? wgrad_tile_kernel<TN, SPW, CT, true> : wgrad_tile_kernel<TN, SPW, CT, false>;
  static bool attr_done[2] = {false, false};  // raise the dynamic-LDS cap once per instantiation
  if (!attr_done[z16]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done[z16] = true;
  }
#ifdef WSR_CT_STAMPS
  a.stamps = getenv("WSR_CT_STAMPS_PTR") ? (unsigned long long*)strtoull(getenv("WSR_CT_STAMPS_PTR"), nullptr, 0) : nullptr;
  a.ablate = WSR_ENV_INT("WSR_CT_ABL", 0);
#endif
  a.prio = WSR_ENV_INT("WSR_CT_PRIO", 0);
  hipLaunchKernelGGL(kern, dim3((unsigned)(combos * a.S)), dim3(WAVES * 64), lds, st, a);
  WSR_LAUNCH_CHECK();
  return 0;
}

}  // namespace

namespace {
int run_tile(WgtArgs& a, int taps, int Cout, int Cin, hipStream_t st);
}

// Returns WSR_EUNSUPPORTED when the shape is outside what the tile kernel covers;
// wsr_conv3d_wgrad then falls back to the per-tap split-K kernel.
// part_stride > 0: deterministic form with n_parts (as returned by a plan call) split copies; plan != nullptr: only
// report the number of spatial splits (*plan) the launch would use.
int wsr_wgrad_tile_bf16(const wsr_conv_t* c, const void* x, const void* dy, float* dw, int tri_base, int tri_step,
                        long part_stride, int n_parts, int* plan, void* stream, const void* x2, int x2_ctot, int x2_c0) {
  if (c->dtype != WSR_BF16 || (c->sx | c->sy | c->sz) != 1) return WSR_EUNSUPPORTED;
  if (x2 && (c->lat || c->upsample_xy || tri_step > 0 || x2_c0 <= 0 || x2_c0 >= c->Cin || x2_c0 % 32 || x2_ctot % 8 ||
             c->Cin - x2_c0 > x2_ctot))
    return WSR_EUNSUPPORTED;
  if (c->lat == 2 && (c->lat_phases || c->lat_mz > 1 || tri_step > 0)) return WSR_EUNSUPPORTED;  // one parity per launch
  if (c->lat == 3 && (c->lat_phases || tri_step > 0 || c->upsample_xy)) return WSR_EUNSUPPORTED;
  const int taps = c->KX * c->KY * c->KZ;
  if (taps > 128) return WSR_EUNSUPPORTED;
  if (c->Cin % 8 || c->in_ctot % 8 || c->in_off % 8 || c->out_ctot % 8 || c->out_off % 8) return WSR_EUNSUPPORTED;
  // the dy DMA moves whole octets of the channel window: it must own them (true for padded NDHWC buffers)
  if (c->out_off + (c->Cout + 7) / 8 * 8 > c->out_ctot) return WSR_EUNSUPPORTED;
  if (c->KX > 8 || c->KY > 8 || c->KZ > 8) return WSR_EUNSUPPORTED;
  WgtArgs a{};
  a.x = (const unsigned short*)x;
  a.x2 = (const unsigned short*)x2; a.x2_ctot = x2_ctot; a.x2_c0 = x2_c0;
  a.dy = (const unsigned short*)dy;
  a.dw = dw;
  a.B = c->B; a.Xi = c->Xi; a.Yi = c->Yi; a.Zi = c->Zi;
  a.Xo = c->Xo; a.Yo = c->Yo; a.Zo = c->Zo;
  a.Cin = c->Cin; a.in_ctot = c->in_ctot; a.in_off = c->in_off;
  a.Cout = c->Cout; a.out_ctot = c->out_ctot; a.out_off = c->out_off;
  a.KX = c->KX; a.KY = c->KY; a.KZ = c->KZ;
  a.px = c->px; a.py = c->py; a.pz = c->pz;
  a.ups = c->upsample_xy ? 1 : 0;
  a.yl_m = c->lat == 2 ? 2 : 1; a.yl_ox = c->lat == 2 ? c->lat_ox : 0; a.yl_oy = c->lat == 2 ? c->lat_oy : 0;
  a.xl_m = a.xl_mz = 1;
  if (c->lat == 3) {  // x on the lattice (2x + ox, 2y + oy, mz*z + oz): Xi, Yi, Zi are the lattice's extents
    a.xl_m = 2; a.xl_ox = c->lat_ox; a.xl_oy = c->lat_oy;
    a.xl_mz = c->lat_mz > 1 ? c->lat_mz : 1; a.xl_oz = c->lat_oz;
  }
  a.tri_base = tri_base; a.tri_step = tri_step;
  a.part_stride = part_stride; a.S_forced = n_parts; a.plan_only = plan ? 1 : 0;
  if (taps == 1 && !c->lat && !c->upsample_xy && c->Zi % 16 != 0 && (c->px | c->py | c->pz) == 0) {
    // A 1x1x1 conv has no halo: its filter gradient is a GEMM over the voxel INDEX, whatever the volume's shape.  The
    // tile picker wants 16-level tiles (the reference's 10-level patches left it no tile that fits two LDS buffers:
    // the cluster configuration's 48 LFF gradients ran on the per-tap kernel, 58 us each against ~25) - so hand it the
    // same voxels as an (n / 128) x 8 x 16 volume.
    const long nv = (long)c->Xi * c->Yi * c->Zi;
    if (nv % 128 == 0 && nv / 128 < (1L << 24)) {
      a.Xi = a.Xo = (int)(nv / 128);
      a.Yi = a.Yo = 8;
      a.Zi = a.Zo = 16;
    }
  }
  hipStream_t st = as_stream(stream);
  const int rc = run_tile(a, taps, c->Cout, c->Cin, st);
  if (plan && rc == 0) *plan = a.S;
  return rc;
}

namespace {
int run_tile(WgtArgs& a, int taps, int Cout, int Cin, hipStream_t st) {
  struct { int Cout, Cin; } cc{Cout, Cin};
  auto* c = &cc;
  if (taps == 1) {  // 1x1x1 (LFF): a plain GEMM over the voxels; 8 slots = 8 c-tiles (128 input channels per chunk)
    if (c->Cout < 64 || c->Cin < 64) return WSR_EUNSUPPORTED;  // tiny GEMMs stay on the per-tap kernel
    return launch_tile<8, 1, 8>(a, st);
  }
  if (taps > 28) {  // 5x5x5: 16 slots per wave, 16 input channels per chunk
    if (c->Cout <= 16) return launch_tile<1, 16, 1>(a, st);
    if (c->Cout % 48 == 0) return launch_tile<3, 16, 1>(a, st);
    return launch_tile<2, 16, 1>(a, st);
  }
  // <= 12 taps (the 2x2x3 parity convs of a sub-pixel up-sampling conv): 6 slots per wave = 12 taps x 4 c-tiles
  if (taps <= 12 && c->Cout >= 64 && c->Cin >= 64) {
    const int rc = launch_tile<4, 6, 4>(a, st);
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  const bool small = a.tri_step == 0 && !WSR_ENV_SET("WSR_WG_NOSMALL");
  // ... with at most 32 output channels (the parity launches of the discriminator's first strided conv, 32 -> 32): 3 slots
  // per wave = 12 taps x 2 c-tiles exactly - on <2,7,2> four of a wave's seven slots were padding
  if (small && taps <= 12 && c->Cout > 16 && c->Cout <= 32) {
    const int rc = launch_tile<2, 3, 2>(a, st);
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  // one 16-channel c-tile, at most 32 output channels (terrain convs 16 -> 16, the discriminator's first conv 3 -> 32):
  // 4 slots per wave = 32 taps x 1 c-tile - the 32-channel chunk of <.,7,2> was half (or three quarters) zeros
  if (small && c->Cin <= 16 && c->Cout <= 32) {
    const int rc = c->Cout <= 16 ? launch_tile<1, 4, 1>(a, st) : launch_tile<2, 4, 1>(a, st);
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  // one 16-channel c-tile against many output channels (the z-folded last conv's gradient with its operands' roles
  // exchanged, engine.py SWAP_THIN_WGRAD: 16 -> 144, 5x5x1): 4 slots per wave = 32 taps x 1 c-tile, 48 output channels
  // per workgroup - the generic choice below would pad the c-chunk to 32 channels and the n-chunk to 64 (37 % useful MFMAs)
  if (c->Cin <= 16 && c->Cout % 48 == 0 && a.tri_step == 0 && !WSR_ENV_SET("WSR_WG_NO341")) {
    const int rc = launch_tile<3, 4, 1>(a, st);
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  // <= 28 taps (3x3x3): 7 slots per wave = 28 taps x 2 c-tiles (32 input channels per chunk)
  if (c->Cout <= 16) return launch_tile<1, 7, 2>(a, st);
  if (c->Cout <= 32) return launch_tile<2, 7, 2>(a, st);
  return launch_tile<4, 7, 2>(a, st);
}
}  // namespace
