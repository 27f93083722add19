// Filter gradient of a stride-1 3-D convolution from LDS-resident spatial tiles (bf16).
//
//   dw[n, tap, c] += sum_v dy[v, n] * x[v + tap - pad, c]
//
// A workgroup owns one (n-chunk of 16*TN output channels, c-chunk of 16*CT input
// channels) pair and walks a strided list of spatial tiles.  Per tile it stages
//   Xs : the input tile WITH its halo, (TX+KX-1)(TY+KY-1)(TZ+KZ-1) voxels x 16*CT ch
//   Ys : the output-gradient tile, TX*TY*TZ voxels x 16*TN ch
// once, and then contracts over the tile's voxels for ALL taps: the x operand of
// tap (kx,ky,kz) is the same LDS image read at a shifted voxel index, so x and dy
// are fetched from HBM/L2 once per tile instead of once per tap.  Accumulators for
// every (tap, 16-channel c-tile) "slot" stay in registers across the whole tile
// list (slots are dealt round-robin to the waves: wavefront-level partial sums),
// and are added to the fp32 gradient once, at the very end.
//
// LDS images are octet-major planes [8-channel octet][voxel][16 B]; the reduction
// index (voxel) is the slow dimension of both MFMA operands, so fragments are
// fetched with the transposing read ds_read_b64_tr_b16.  The plane stride is
// == 64 (mod 256) bytes, which makes the four 64-byte runs a half-wave touches
// (2 voxel quads x 2 octets) land on disjoint banks.
//
// `tri_step` > 0 describes the block-triangular structure of a residual dense
// block: output channel n belongs to conv i = n / tri_step whose input is only
// channels [0, tri_base + i*tri_step) of the shared dense buffer, so the four
// growth convs of an RDB (reference torch_blocks.py:256-267) are ONE launch.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef s16x4_t __attribute__((address_space(3))) * lds_s16x4_ptr;

struct WgtArgs {
  const unsigned short* x;
  const unsigned short* dy;
  float* dw;
  int B, Xi, Yi, Zi, Xo, Yo, Zo;
  int Cin, in_ctot, in_off;    // Cin = padded channel count of the x window (dw row length)
  int Cout, out_ctot, out_off; // Cout = channels of dy that are real (dw rows)
  int KX, KY, KZ, px, py, pz, ups;
  int TX, TY, TZ;              // output tile
  int CT;                      // 16-channel c-tiles per c-chunk
  int n_chunks, c_chunks, S;   // grid = n_chunks * c_chunks * S
  int tiles_x, tiles_y, tiles_z, ntiles;
  int PX, PY;                  // plane strides (bytes)
  int off_mtab, off_htab, off_vtab, off_xs, off_ys;  // LDS carve (bytes)
  int tri_base, tri_step;
};

__device__ __forceinline__ uint4 tr_frag(const char* lo, const char* hi) {
  s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lo));
  s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(hi));
  uint2 l2 = __builtin_bit_cast(uint2, a), h2 = __builtin_bit_cast(uint2, b);
  return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

template <int WAVES, int TN, int SPW>
__global__ __launch_bounds__(WAVES * 64) void wgrad_tile_kernel(const WgtArgs a) {
  constexpr int NT = WAVES * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

  const int Ly = a.TY + a.KY - 1, Lz = a.TZ + a.KZ - 1, Lx = a.TX + a.KX - 1;
  const int L = Lx * Ly * Lz;
  const int M = a.TX * a.TY * a.TZ;
  const int taps = a.KX * a.KY * a.KZ;

  unsigned* mtab = reinterpret_cast<unsigned*>(smem + a.off_mtab);          // [M]  ox | oy<<8 | oz<<16
  unsigned short* htab = reinterpret_cast<unsigned short*>(smem + a.off_htab);  // [M] halo index of voxel m
  unsigned* vtab = reinterpret_cast<unsigned*>(smem + a.off_vtab);          // [L]  hx | hy<<8 | hz<<16
  char* Xs = smem + a.off_xs;
  char* Ys = smem + a.off_ys;

  // ---- which chunk pair / spatial slice ------------------------------------------
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int cc = bid % a.c_chunks;
  bid /= a.c_chunks;
  const int nc = bid % a.n_chunks;
  const int s0 = bid / a.n_chunks;
  const int c0 = cc * 16 * a.CT, n0 = nc * 16 * TN;

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

  // ---- per-kernel tables ------------------------------------------------------------
  for (int m = t; m < M; m += NT) {
    const int oz = m % a.TZ, r = m / a.TZ;
    const int oy = r % a.TY, ox = r / a.TY;
    mtab[m] = ox | (oy << 8) | (oz << 16);
    htab[m] = (unsigned short)((ox * Ly + oy) * Lz + oz);
  }
  for (int v = t; v < L; v += NT) {
    const int hz = v % Lz, r = v / Lz;
    const int hy = r % Ly, hx = r / Ly;
    vtab[v] = hx | (hy << 8) | (hz << 16);
  }

  // ---- this wave's slots: (tap, c-tile) pairs ------------------------------------------
  const int nslots = taps * a.CT;
  int soff[SPW];  // byte offset of the slot's x image relative to the un-shifted one
#pragma unroll
  for (int j = 0; j < SPW; ++j) {
    const int sj = wave + WAVES * j;
    int off = 0;
    if (sj < nslots) {
      const int tap = sj / a.CT, ct = sj % a.CT;
      const int kz = tap % a.KZ, r = tap / a.KZ;
      const int ky = r % a.KY, kx = r / a.KY;
      off = 2 * ct * a.PX + ((kx * Ly + ky) * Lz + kz) * 16;
    }
    soff[j] = __builtin_amdgcn_readfirstlane(off);
  }

  f32x4_t acc[SPW][TN];
#pragma unroll
  for (int j = 0; j < SPW; ++j)
#pragma unroll
    for (int i = 0; i < TN; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // tr-read lane roles: 16-lane group G supplies rows (voxels) 8G+q, columns 4p..4p+3
  const int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int lane_x = (p >> 1) * a.PX + (p & 1) * 8;
  const int lane_y = (p >> 1) * a.PY + (p & 1) * 8;
  const int XP = 2 * a.CT, YP = 2 * TN;
  const int U = a.ups ? 1 : 0;
  const int ksteps = M >> 5;

  for (int tile = s0; tile < a.ntiles; tile += a.S) {
    int r = tile;
    const int tz = r % a.tiles_z; r /= a.tiles_z;
    const int ty = r % a.tiles_y; r /= a.tiles_y;
    const int tx = r % a.tiles_x;
    const int b = r / a.tiles_x;
    const int x0 = tx * a.TX, y0 = ty * a.TY, z0 = tz * a.TZ;

    __syncthreads();  // tables ready / previous tile's fragment reads done
    // ---- stage x (with halo) ------------------------------------------------------
    for (int i = t; i < L * XP; i += NT) {
      const int v = i / XP, pl = i - v * XP;
      const unsigned hv = vtab[v];
      const int gx = x0 - a.px + (int)(hv & 255), gy = y0 - a.py + (int)((hv >> 8) & 255),
                gz = z0 - a.pz + (int)(hv >> 16);
      uint4 val = make_uint4(0, 0, 0, 0);
      const int c = c0 + 8 * pl;
      if ((unsigned)gx < (unsigned)(a.Xi << U) && (unsigned)gy < (unsigned)(a.Yi << U) &&
          (unsigned)gz < (unsigned)a.Zi && c < a.Cin) {
        const long vox = (((long)b * a.Xi + (gx >> U)) * a.Yi + (gy >> U)) * a.Zi + gz;
        val = *reinterpret_cast<const uint4*>(a.x + vox * a.in_ctot + a.in_off + c);
      }
      *reinterpret_cast<uint4*>(Xs + pl * a.PX + v * 16) = val;
    }
    // ---- stage dy -------------------------------------------------------------------
    for (int i = t; i < M * YP; i += NT) {
      const int m = i / YP, pl = i - m * YP;
      const unsigned mv = mtab[m];
      const int gx = x0 + (int)(mv & 255), gy = y0 + (int)((mv >> 8) & 255), gz = z0 + (int)(mv >> 16);
      uint4 val = make_uint4(0, 0, 0, 0);
      const int n = n0 + 8 * pl;
      if (gx < a.Xo && gy < a.Yo && gz < a.Zo && n < a.Cout) {
        const long vox = (((long)b * a.Xo + gx) * a.Yo + gy) * a.Zo + gz;
        const unsigned short* src = a.dy + vox * a.out_ctot + a.out_off + n;
        if (n + 8 <= a.Cout) {
          val = *reinterpret_cast<const uint4*>(src);
        } else {  // ragged tail of the channel window (e.g. the 3-channel SR output)
          unsigned short tmp[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) tmp[e] = (n + e < a.Cout) ? src[e] : (unsigned short)0;
          val = *reinterpret_cast<uint4*>(tmp);
        }
      }
      *reinterpret_cast<uint4*>(Ys + pl * a.PY + m * 16) = val;
    }
    __syncthreads();

    // ---- contract over the tile's voxels, 32 per step --------------------------------------
    for (int ks = 0; ks < ksteps; ++ks) {
      const int m_lo = ks * 32 + 8 * G + q;
      const int h_lo = htab[m_lo], h_hi = htab[m_lo + 4];
      uint4 af[TN];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const char* base = Ys + 2 * i * a.PY + lane_y;
        af[i] = tr_frag(base + m_lo * 16, base + (m_lo + 4) * 16);
      }
      const char* xlo = Xs + lane_x + h_lo * 16;
      const char* xhi = Xs + lane_x + h_hi * 16;
#pragma unroll
      for (int j = 0; j < SPW; ++j) {
        if (wave + WAVES * j < nslots) {
          const uint4 bf = tr_frag(xlo + soff[j], xhi + soff[j]);
#pragma unroll
          for (int i = 0; i < TN; ++i)
            if (act[i])
              acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[i]),
                                                                  __builtin_bit_cast(bf16x8_t, bf), acc[j][i], 0, 0, 0);
        }
      }
    }
  }

  // ---- add this workgroup's partial sums: acc[j][i][r] -> n = n0+16i+4G+r, c = c0+16ct+(lane&15)
#pragma unroll
  for (int j = 0; j < SPW; ++j) {
    const int sj = wave + WAVES * j;
    if (sj >= nslots) continue;
    const int tap = sj / a.CT, ct = sj % a.CT;
    const int c = c0 + 16 * ct + (lane & 15);
    if (c >= a.Cin) continue;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      if (!act[i]) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 16 * i + 4 * G + r;
        if (n < a.Cout) atomicAdd(a.dw + ((long)n * taps + tap) * a.Cin + c, acc[j][i][r]);
      }
    }
  }
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

template <int WAVES, int TN, int SPW>
int launch_tile(WgtArgs& a, int wg_target, hipStream_t st) {
  const int taps = a.KX * a.KY * a.KZ;
  const int M = a.TX * a.TY * a.TZ;
  const int L = (a.TX + a.KX - 1) * (a.TY + a.KY - 1) * (a.TZ + a.KZ - 1);
  if (taps * a.CT > WAVES * SPW || (M & 31) || L > 65535) return WSR_EUNSUPPORTED;
  a.PX = round_up(L * 16, 256) + 64;
  a.PY = round_up(M * 16, 256) + 64;
  a.off_mtab = 0;
  a.off_htab = a.off_mtab + M * 4;
  a.off_vtab = round_up(a.off_htab + M * 2, 16);
  a.off_xs = round_up(a.off_vtab + L * 4, 256);
  a.off_ys = a.off_xs + 2 * a.CT * a.PX;
  const size_t lds = (size_t)a.off_ys + (size_t)2 * TN * a.PY;
  if (lds > 160 * 1024) return WSR_EUNSUPPORTED;
  a.n_chunks = (a.Cout + 16 * TN - 1) / (16 * TN);
  a.c_chunks = (a.Cin + 16 * a.CT - 1) / (16 * a.CT);
  a.tiles_x = (a.Xo + a.TX - 1) / a.TX;
  a.tiles_y = (a.Yo + a.TY - 1) / a.TY;
  a.tiles_z = (a.Zo + a.TZ - 1) / a.TZ;
  a.ntiles = a.B * a.tiles_x * a.tiles_y * a.tiles_z;
  const int combos = a.n_chunks * a.c_chunks;
  int S = (wg_target + combos - 1) / combos;
  if (S > a.ntiles) S = a.ntiles;
  if (S < 1) S = 1;
  a.S = S;
  auto kern = wgrad_tile_kernel<WAVES, TN, SPW>;
  static bool attr_done = false;  // raise the dynamic-LDS cap once per instantiation
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(combos * S)), dim3(WAVES * 64), lds, st, a);
  WSR_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// Returns WSR_EUNSUPPORTED when the shape is outside what the tile kernel covers;
// wsr_conv3d_wgrad then falls back to the per-tap split-K kernel.
int wsr_wgrad_tile_bf16(const wsr_conv_t* c, const void* x, const void* dy, float* dw, int tri_base, int tri_step,
                        void* stream) {
  if (c->dtype != WSR_BF16 || (c->sx | c->sy | c->sz) != 1) return WSR_EUNSUPPORTED;
  const int taps = c->KX * c->KY * c->KZ;
  if (taps < 2) return WSR_EUNSUPPORTED;
  if (c->Cin % 8 || c->in_ctot % 8 || c->in_off % 8 || c->out_ctot % 8 || c->out_off % 8) return WSR_EUNSUPPORTED;
  const int ux = c->upsample_xy ? 2 : 1;
  WgtArgs a{};
  a.x = (const unsigned short*)x;
  a.dy = (const unsigned short*)dy;
  a.dw = dw;
  a.B = c->B; a.Xi = c->Xi; a.Yi = c->Yi; a.Zi = c->Zi;
  a.Xo = c->Xo; a.Yo = c->Yo; a.Zo = c->Zo;
  a.Cin = c->Cin; a.in_ctot = c->in_ctot; a.in_off = c->in_off;
  a.Cout = c->Cout; a.out_ctot = c->out_ctot; a.out_off = c->out_off;
  a.KX = c->KX; a.KY = c->KY; a.KZ = c->KZ;
  a.px = c->px; a.py = c->py; a.pz = c->pz;
  a.ups = ux == 2;
  a.tri_base = tri_base; a.tri_step = tri_step;
  if (c->KX > 8 || c->KY > 8 || c->KZ > 8) return WSR_EUNSUPPORTED;
  a.TZ = c->Zo <= 12 ? c->Zo : 8;
  a.CT = 1;
  hipStream_t st = as_stream(stream);
  if (taps > 28) {  // 5x5x5: eight waves, 16 slots each, one workgroup per CU
    if (taps > 128) return WSR_EUNSUPPORTED;
    a.TX = 8; a.TY = 8;
    if (a.TZ > 10) a.TZ = 8;
    if (c->Cout <= 16) return launch_tile<8, 1, 16>(a, 256 * 3, st);
    if (c->Cout % 48 == 0) return launch_tile<8, 3, 16>(a, 256 * 3, st);
    return launch_tile<8, 2, 16>(a, 256 * 3, st);
  }
  // <= 28 taps (3x3x3): four waves, 7 slots each, several workgroups per CU overlap load and MFMA
  a.TX = 4; a.TY = 8;
  if (c->Cout <= 16) return launch_tile<4, 1, 7>(a, 256 * 8, st);
  if (c->Cout <= 32) return launch_tile<4, 2, 7>(a, 256 * 8, st);
  return launch_tile<4, 4, 7>(a, 256 * 8, st);
}
