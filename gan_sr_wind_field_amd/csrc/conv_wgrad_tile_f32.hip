// Filter gradient of a stride-1 3-D convolution from LDS-resident tiles, fp32 (the reference's own arithmetic).
//
//   dw[n, tap, c] = sum_v dy[v, n] * x[v + tap - pad, c]
//
// The generic per-tap kernel (conv_wgrad.hip) gives every (tap, 48 x 48 channel tile) its own workgroups, which
// re-read dy and x once per tap through L2: at the 64^3 -> 256^3 generator-only configuration (BASELINE.json
// configs[1]) the 5x5x5 144 -> 144 gradient ran at 52 TFLOP/s, a third of the fp32 MFMA rate, and the filter
// gradients were 45 % of the step.  Here a workgroup (8 waves) owns 16*TN output channels x 16*CT input channels and
// ALL taps: every (tap, c-tile) "slot" has its accumulators in registers (slots dealt round-robin to the waves),
// a small spatial tile of dy and of x with its halo is staged in LDS once (double-buffered LDS-DMA) and re-read at
// a shifted row for every tap.  v_mfma_f32_16x16x4_f32 takes its reduction index across the four 16-lane groups of
// the wave, so lane (r, g) simply reads channel r of voxel v0 + g - no transposing read is needed as in bf16.
// The sum over a K-step is an fmaf chain in voxel order (exact fp32, bit-reproducible); the spatial splits store to
// their own copies (wsr_conv3d_wgrad_parts) or add with float atomics, exactly as the bf16 tile kernel.
#include <cstdlib>

#include "common.h"

namespace {

struct WgfArgs {
  const float* x;
  const float* x2;              // input channels >= x2_c0 come from channels [0, ...) of this tensor (x2_ctot per voxel), or NULL
  int x2_ctot, x2_c0;
  const float* dy;
  float* dw;
  const void* zero16;
  int B, Xi, Yi, Zi, Xo, Yo, Zo;
  int Cin, in_ctot, in_off;     // Cin = channels of the x window (dw row length, multiple of 4)
  int Cout, out_ctot, out_off;
  int KX, KY, KZ, px, py, pz, ups;
  int TX, TY, TZ;               // output tile; TZ % 4 == 0
  int n_chunks, c_chunks, S;    // grid = n_chunks * c_chunks * S
  int tiles_x, tiles_y, tiles_z, ntiles;
  int xp_bytes, yp_bytes;       // one c-tile plane of the x image / one n-tile plane of the dy image
  int buf_bytes;                // x image + dy image of one buffer
  long part_stride;             // > 0: split s STORES its sums at dw + s*part_stride; 0: float atomics into dw
};

__device__ __forceinline__ void wgf_glds16(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_addr)
      : "memory");
}
__device__ __forceinline__ void wgf_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int TN, int SPW, int CT>
__global__ __launch_bounds__(512) void wgrad_tile_f32_kernel(const WgfArgs a) {
  constexpr int WAVES = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int Ly = a.TY + a.KY - 1, Lz = a.TZ + a.KZ - 1, Lx = a.TX + a.KX - 1;
  const int L = Lx * Ly * Lz;
  const int M = a.TX * a.TY * a.TZ;
  const int taps = a.KX * a.KY * a.KZ;
  typedef __attribute__((address_space(3))) char* lptr_t;
  const unsigned buf_lds = (unsigned)(unsigned long)(lptr_t)smem;

  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int cc = bid % a.c_chunks; bid /= a.c_chunks;
  const int nc = bid % a.n_chunks;
  const int s0 = bid / a.n_chunks;
  const int c0 = cc * 16 * CT, n0 = nc * 16 * TN;
  const int U = a.ups ? 1 : 0;
  // the tensor this workgroup's c-chunk lives in (wsr_conv3d_wgrad_parts_x2: the generator's concat as two tensors)
  const bool second = a.x2 != nullptr && c0 >= a.x2_c0;
  const float* xt = second ? a.x2 : a.x;
  const int x_ctot = second ? a.x2_ctot : a.in_ctot;
  const int x_off = second ? c0 - a.x2_c0 : a.in_off + c0;

  // ---- DMA of one tile: x image [c-tile][halo voxel][64 B], dy image [n-tile][voxel][64 B] -----------------------
  // 1 KB units of 16 rows x 4 pieces; the geometry is resolved per tile (a tile is tens of microseconds of fp32
  // matrix work: the address arithmetic is noise)
  const int XUP = (L + 15) >> 4, YUP = (M + 15) >> 4;  // units per plane
  const unsigned short* zsrc = reinterpret_cast<const unsigned short*>(a.zero16);
  auto issue_tile = [&](int tile, int buf) {
    int r = tile;
    const int tz = r % a.tiles_z; r /= a.tiles_z;
    const int ty = r % a.tiles_y; r /= a.tiles_y;
    const int tx = r % a.tiles_x;
    const int b = r / a.tiles_x;
    const int x0 = tx * a.TX, y0 = ty * a.TY, z0 = tz * a.TZ;
    const unsigned dstx = buf_lds + buf * a.buf_bytes;
    const unsigned dsty = dstx + CT * a.xp_bytes;
    const int row = lane >> 2, pc = lane & 3;  // row of the unit, 16-byte piece (4 channels) of the row
    for (int u = wave; u < XUP * CT; u += WAVES) {
      const int ct = u / XUP, h = (u - ct * XUP) * 16 + row;
      const void* src = zsrc;
      const int c = c0 + 16 * ct + 4 * pc;
      if (h < L && c < a.Cin) {
        const int hz = h % Lz, qq = h / Lz;
        const int hy = qq % Ly, hx = qq / Ly;
        const int gx = x0 - a.px + hx, gy = y0 - a.py + hy, gz = z0 - a.pz + hz;
        if ((unsigned)gx < (unsigned)(a.Xi << U) && (unsigned)gy < (unsigned)(a.Yi << U) && (unsigned)gz < (unsigned)a.Zi)
          src = xt + ((((long)b * a.Xi + (gx >> U)) * a.Yi + (gy >> U)) * a.Zi + gz) * x_ctot + x_off + 16 * ct + 4 * pc;
      }
      wgf_glds16(src, __builtin_amdgcn_readfirstlane(dstx + ct * a.xp_bytes + (u - ct * XUP) * 1024));
    }
    for (int u = wave; u < YUP * TN; u += WAVES) {
      const int nt = u / YUP, v = (u - nt * YUP) * 16 + row;
      const void* src = zsrc;
      const int n = n0 + 16 * nt + 4 * pc;
      if (v < M && n < a.Cout) {  // (channel windows are whole 16-byte pieces: the host checks)
        const int oz = v % a.TZ, qq = v / a.TZ;
        const int oy = qq % a.TY, ox = qq / a.TY;
        const int gx = x0 + ox, gy = y0 + oy, gz = z0 + oz;
        if (gx < a.Xo && gy < a.Yo && gz < a.Zo)
          src = a.dy + ((((long)b * a.Xo + gx) * a.Yo + gy) * a.Zo + gz) * a.out_ctot + a.out_off + n;
      }
      wgf_glds16(src, __builtin_amdgcn_readfirstlane(dsty + nt * a.yp_bytes + (u - nt * YUP) * 1024));
    }
  };

  // ---- this wave's slots: (tap, c-tile) pairs; byte offset of the slot's x rows from the un-shifted tap ----------
  const int nslots = taps * CT;
  int soff[SPW];
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
    soff[j] = __builtin_amdgcn_readfirstlane(off * 64 + ct * a.xp_bytes);
  }
  f32x4_t acc[SPW][TN];
#pragma unroll
  for (int j = 0; j < SPW; ++j)
#pragma unroll
    for (int i = 0; i < TN; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // K-step ks contracts voxels 4 ks .. 4 ks + 3 of the tile (z-consecutive: TZ % 4 == 0); lane group g takes voxel
  // 4 ks + g: row (ox*Ly + oy)*Lz + oz of the halo image, row 4 ks + g of the dy image
  const int ksteps = M >> 2;
  const int lane_x = fg * 64 + fr * 4, lane_y = fg * 64 + fr * 4;

  issue_tile(s0, 0);
  for (int it = 0;; ++it) {
    wgf_dma_wait();
    __syncthreads();  // tile `it` has landed for everybody; the other buffer is free
    const int nxt = s0 + (it + 1) * a.S;
    if (nxt < a.ntiles) issue_tile(nxt, (it + 1) & 1);
    const char* Xs = smem + (it & 1) * a.buf_bytes;
    const char* Ys = Xs + CT * a.xp_bytes;
    // Operand pipeline: the dy values of K-step ks + 1 and its first x value are requested while the MFMAs of K-step
    // ks run (a K-step of the 3x3x3 instantiation is only 14 MFMAs = 448 matrix-pipe cycles: an LDS round trip in
    // front of every one of them cost a quarter of the launch)
    int ox = 0, oy = 0, oz = 0;  // tile coordinates of voxel 4 ks
    const char* xr = Xs + lane_x;
    float afn[TN], bfn;
#pragma unroll
    for (int i = 0; i < TN; ++i) afn[i] = *reinterpret_cast<const float*>(Ys + lane_y + i * a.yp_bytes);
    bfn = *reinterpret_cast<const float*>(xr + soff[0]);
    for (int ks = 0; ks < ksteps; ++ks) {
      float af[TN], bf[2];
#pragma unroll
      for (int i = 0; i < TN; ++i) af[i] = afn[i];
      bf[0] = bfn;
      // coordinates / rows of the next K-step (the last one re-reads itself: no branch around the requests)
      oz += 4;
      if (oz >= a.TZ) { oz = 0; if (++oy >= a.TY) { oy = 0; ++ox; } }
      const bool last = ks + 1 == ksteps;
      const char* xn = last ? xr : Xs + ((ox * Ly + oy) * Lz + oz) * 64 + lane_x;
      const char* yn = Ys + (last ? ks : ks + 1) * 256 + lane_y;
#pragma unroll
      for (int i = 0; i < TN; ++i) afn[i] = *reinterpret_cast<const float*>(yn + i * a.yp_bytes);
      // (the scheduler is fenced: left alone it sinks every x request to just in front of its MFMAs - "read, wait,
      // 3 MFMAs" per slot on ONE register - and each slot pays an LDS round trip)
#pragma unroll
      for (int j = 0; j < SPW; ++j) {
        if (j + 1 < SPW) bf[(j + 1) & 1] = *reinterpret_cast<const float*>(xr + soff[j + 1]);
        else bfn = *reinterpret_cast<const float*>(xn + soff[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TN; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j & 1], acc[j][i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      xr = xn;
    }
    if (nxt >= a.ntiles) break;
  }

  // ---- acc[j][i][r] -> n = n0 + 16 i + 4 fg + r, c = c0 + 16 ct + fr ------------------------------------------------
  float* const dwp = a.dw + (long)s0 * a.part_stride;
  const bool store = a.part_stride > 0;
#pragma unroll
  for (int j = 0; j < SPW; ++j) {
    const int sj = wave + WAVES * j;
    if (sj >= nslots) continue;
    const int tap = sj / CT, ct = sj % CT;
    const int c = c0 + 16 * ct + fr;
    if (c >= a.Cin) continue;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 16 * i + 4 * fg + r;
        if (n < a.Cout) {
          float* q = dwp + ((long)n * taps + tap) * a.Cin + c;
          if (store) *q = acc[j][i][r];
          else atomicAdd(q, acc[j][i][r]);
        }
      }
  }
}

__device__ uint4 g_wgf_zero16 = {0u, 0u, 0u, 0u};

template <int TN, int SPW, int CT>
int launch_wgf(WgfArgs& a, int n_parts, int* plan, hipStream_t st) {
  const int taps = a.KX * a.KY * a.KZ;
  if (taps * CT > SPW * 8) return WSR_EUNSUPPORTED;
  if (!plan) {
    static void* zp = nullptr;
    if (!zp) {
      hipError_t e = hipGetSymbolAddress(&zp, HIP_SYMBOL(g_wgf_zero16));
      if (e != hipSuccess) return (int)e;
    }
    a.zero16 = zp;
  }
  // spatial tile: z runs of up to 16 voxels (a multiple of 4), a few voxels in x and y - the halo image of 16*CT fp32
  // channels must fit twice next to the dy image
  a.TZ = a.Zo >= 16 ? 16 : (a.Zo + 3) / 4 * 4;
  a.TX = 4; a.TY = 4;
  for (;;) {
    const int L = (a.TX + a.KX - 1) * (a.TY + a.KY - 1) * (a.TZ + a.KZ - 1);
    const int M = a.TX * a.TY * a.TZ;
    a.xp_bytes = (L + 15) / 16 * 1024;
    a.yp_bytes = (M + 15) / 16 * 1024;
    a.buf_bytes = CT * a.xp_bytes + TN * a.yp_bytes;
    if (2 * a.buf_bytes <= 160 * 1024) break;
    if (a.TY >= a.TX && a.TY > 1) a.TY >>= 1;
    else if (a.TX > 1) a.TX >>= 1;
    else if (a.TZ > 4) a.TZ -= 4;
    else return WSR_EUNSUPPORTED;
  }
  a.tiles_x = (a.Xo + a.TX - 1) / a.TX;
  a.tiles_y = (a.Yo + a.TY - 1) / a.TY;
  a.tiles_z = (a.Zo + a.TZ - 1) / a.TZ;
  const long ntiles = (long)a.B * a.tiles_x * a.tiles_y * a.tiles_z;
  if (ntiles >= (1l << 31)) return WSR_EUNSUPPORTED;
  a.ntiles = (int)ntiles;
  a.n_chunks = (a.Cout + 16 * TN - 1) / (16 * TN);
  a.c_chunks = (a.Cin + 16 * CT - 1) / (16 * CT);
  const int combos = a.n_chunks * a.c_chunks;
  int S = 256 / combos;  // ONE round of workgroups over the 256 CUs (a 257th workgroup would double the launch time)
  if (S > a.ntiles) S = a.ntiles;
  if (S < 1) S = 1;
  a.S = S;
  if (plan) { *plan = S; return 0; }
  if (a.part_stride > 0 && n_parts != S) return WSR_EINVAL;  // the caller sized `parts` for another split
  auto kern = wgrad_tile_f32_kernel<TN, SPW, CT>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(combos * S)), dim3(512), (size_t)2 * a.buf_bytes, st, a);
  WSR_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// WSR_EUNSUPPORTED: shape outside this kernel (strided / lattice convs, ragged channel windows) - the caller falls back
// to the per-tap kernel.  part_stride / n_parts / plan as in wsr_wgrad_tile_bf16.
int wsr_wgrad_tile_f32(const wsr_conv_t* c, const void* x, const void* dy, float* dw, long part_stride, int n_parts,
                       int* plan, void* stream, const void* x2, int x2_ctot, int x2_c0) {
  if (c->dtype != WSR_F32 || (c->sx | c->sy | c->sz) != 1 || c->lat) return WSR_EUNSUPPORTED;
  if (x2 && (c->upsample_xy || x2_c0 <= 0 || x2_c0 >= c->Cin || x2_c0 % 128 || x2_ctot % 4 || c->Cin - x2_c0 > x2_ctot))
    return WSR_EUNSUPPORTED;  // (x2_c0 a multiple of every instantiation's c-chunk: 16 .. 128 channels)
  const int taps = c->KX * c->KY * c->KZ;
  if (taps > 128) return WSR_EUNSUPPORTED;
  if (c->Cin % 4 || c->in_ctot % 4 || c->in_off % 4 || c->out_ctot % 4 || c->out_off % 4) return WSR_EUNSUPPORTED;
  if (c->out_off + (c->Cout + 3) / 4 * 4 > c->out_ctot) return WSR_EUNSUPPORTED;  // the dy DMA moves whole pieces
  if (WSR_ENV_SET("WSR_NO_WGRAD_F32_TILE")) return WSR_EUNSUPPORTED;  // tuning / A-B switch
  WgfArgs a{};
  a.x = (const float*)x; a.dy = (const float*)dy; a.dw = dw;
  a.x2 = (const float*)x2; a.x2_ctot = x2_ctot; a.x2_c0 = x2_c0;
  a.B = c->B; a.Xi = c->Xi; a.Yi = c->Yi; a.Zi = c->Zi; a.Xo = c->Xo; a.Yo = c->Yo; a.Zo = c->Zo;
  a.Cin = c->Cin; a.in_ctot = c->in_ctot; a.in_off = c->in_off;
  a.Cout = c->Cout; a.out_ctot = c->out_ctot; a.out_off = c->out_off;
  a.KX = c->KX; a.KY = c->KY; a.KZ = c->KZ; a.px = c->px; a.py = c->py; a.pz = c->pz;
  a.ups = c->upsample_xy ? 1 : 0;
  a.part_stride = part_stride;
  hipStream_t st = as_stream(stream);
  if (taps == 1) {  // 1x1x1 (LFF): 8 slots = 8 c-tiles
    if (c->Cout < 64 || c->Cin < 64) return WSR_EUNSUPPORTED;
    return launch_wgf<4, 1, 8>(a, n_parts, plan, st);
  }
  if (taps > 28) {  // 5x5x5
    if (c->Cout <= 16) return launch_wgf<1, 16, 1>(a, n_parts, plan, st);
    if (c->Cout % 48 == 0) return launch_wgf<3, 16, 1>(a, n_parts, plan, st);
    return launch_wgf<2, 16, 1>(a, n_parts, plan, st);
  }
  if (c->Cin <= 16) {  // few input channels (feature conv, terrain convs): one c-tile, 4 slots per wave
    if (c->Cout <= 16) return launch_wgf<1, 4, 1>(a, n_parts, plan, st);
    return launch_wgf<4, 4, 1>(a, n_parts, plan, st);
  }
  if (c->Cout <= 16) return launch_wgf<1, 7, 2>(a, n_parts, plan, st);
  if (c->Cout <= 32) return launch_wgf<2, 7, 2>(a, n_parts, plan, st);
  return launch_wgf<4, 7, 2>(a, n_parts, plan, st);
}
