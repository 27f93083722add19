// (implementation header shared by the conv_tile_*.hip translation units)
//
// Stride-1 3-D convolution (forward and input-gradient) from an LDS-resident halo tile, bf16.
//
//   y[v, n] = sum_{tap, c} x[v - pad + tap, c] * w[n, tap, c]
//
// A workgroup owns one spatial output tile (TX x TY x TZ voxels) and a group of
// 16-wide output-channel tiles.  The reduction channels are walked in chunks of
// CK = 32/TPK channels; per chunk the input tile WITH its halo is staged in LDS
// once and then re-read at a shifted voxel index for every filter tap, so each
// activation byte leaves L2/HBM once per chunk instead of once per tap (the 5x5x5
// 144->144 conv re-uses a staged voxel 125 times).  One MFMA K-step (32) covers
// TPK taps x CK channels:
//   TPK = 1 : 32 channels of one tap        (1x1x1 convs)
//   TPK = 2 : 16 channels of a tap pair     (channel counts that are multiples of 16)
//   TPK = 4 :  8 channels of four taps      (1/3/4-channel tensors padded to 8)
// Filters are pre-packed in MFMA-fragment order (wsr_pack_filter_frag), streamed
// through a double-buffered LDS ring one "stage" (a few K-steps) ahead of the
// MFMAs, and read back conflict-free as linear 1 KB fragments.
//
// LDS activation image: octet-major planes [8-channel octet][halo voxel][16 B],
// plane stride == 0 (mod 256) so the 16 voxel rows of a fragment, contiguous along
// z, are bank-conflict free for ds_read_b128 at every tap shift.
//
// The MFMA is issued as D = W * X^T: a lane ends up with 4 consecutive output
// channels of one voxel (8-byte vector stores into the NDHWC channel window).
// The input-gradient pass is the same kernel over dy with the transposed,
// tap-flipped filter and pad' = K-1-pad.
#pragma once
#include <cstdlib>

#include "common.h"

struct CtArgs {
  const unsigned short* in;
  const unsigned short* wf;  // fragment-packed filter: [chunk][kstep][ntile][64 lanes][8]
  void* out;
  const void* zero16;        // 16 zero bytes in global memory (source of out-of-range DMA lanes)
  const float* bias;
  const float* chan_scale;
  const unsigned short* res;
  int res_ctot, res_off;
  float alpha, beta, slope;
  int act, out_planar;
  int B, Xi, Yi, Zi, Xo, Yo, Zo, ups;
  int in_ctot, in_off, nchunks;   // reduction channels = nchunks * CK, window [in_off, ...)
  int cin_valid;                  // channels of the window that exist in memory (multiple of 8)
  int Cout, out_ctot, out_off;
  int KX, KY, KZ, px, py, pz;
  int TX, TY, TZ;
  int tiles_x, tiles_y, tiles_z, ntiles;
  int nts;          // K-steps per chunk = ceil(taps / TPK)
  int TS;           // K-steps per weight stage
  int NT_total;     // 16-wide output-channel tiles
  int ngroups;      // n-tile groups (grid = ntiles * ngroups)
  int P;            // activation plane stride (bytes)
  int off_mtab, off_htab, off_vtab, off_ttab, off_xs, off_ws;
  int vec_ok;
  const unsigned short* mask_y;  // LeakyReLU-backward mask source (saved forward output) or NULL
  int mask_ctot, mask_off, mask_c0, mask_c1;
  float mask_slope;
  int xbufs;        // activation buffers in LDS: 2 = next chunk prefetched during the MFMAs
};

namespace {

// LDS-DMA of 16 B per lane: LDS[lds_addr + 16*lane] <- *gsrc.  Issued as inline asm so that hipcc does
// not serialise it against the LDS fragment reads of the phase in flight (with the builtin it puts
// s_waitcnt vmcnt(0) in front of every ds_read that follows); completion is waited for explicitly
// (dma_wait) before the barrier that publishes the buffer.  M0 carries the wave-uniform LDS base
// and is restored, as the compiler reserves it.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_addr)
      : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int WM, int WN, int TM, int TN, int TPK, bool PIPE, bool MASK>
__global__ __launch_bounds__(WM * WN * 64) void conv_tile_kernel(const CtArgs a) {
  constexpr int WAVES = WM * WN, NT = WAVES * 64;
  constexpr int PL = 4 / TPK;      // octet planes per chunk
  constexpr int CK = 8 * PL;       // channels per chunk
  constexpr int NTW = WN * TN;     // n-tiles per workgroup
  constexpr bool STAGGER = true;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int Lx = a.TX + a.KX - 1, Ly = a.TY + a.KY - 1, Lz = a.TZ + a.KZ - 1;
  const int L = Lx * Ly * Lz;
  const int M = a.TX * a.TY * a.TZ;  // <= MR
  constexpr int MR = WM * TM * 16;    // MFMA rows of the workgroup
  const int taps = a.KX * a.KY * a.KZ;

  unsigned* mtab = reinterpret_cast<unsigned*>(smem + a.off_mtab);
  unsigned short* htab = reinterpret_cast<unsigned short*>(smem + a.off_htab);
  unsigned* vtab = reinterpret_cast<unsigned*>(smem + a.off_vtab);
  int* ttab = reinterpret_cast<int*>(smem + a.off_ttab);
  char* Xs = smem + a.off_xs;
  char* Ws = smem + a.off_ws;

  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int ng = bid % a.ngroups;
  const int tile = bid / a.ngroups;
  int r = tile;
  const int tz = r % a.tiles_z; r /= a.tiles_z;
  const int ty = r % a.tiles_y; r /= a.tiles_y;
  const int tx = r % a.tiles_x;
  const int b = r / a.tiles_x;
  const int x0 = tx * a.TX, y0 = ty * a.TY, z0 = tz * a.TZ;
  const int nt0 = ng * NTW;  // first n-tile of this workgroup

  // ---- tables -----------------------------------------------------------------------
  for (int m = t; m < MR; m += NT) {  // rows >= M (tile volume) are padding: flagged, read voxel 0
    const int oz = m % a.TZ, q = m / a.TZ;
    const int oy = q % a.TY, ox = q / a.TY;
    mtab[m] = m < M ? (ox | (oy << 8) | (oz << 16)) : (1u << 24);
    htab[m] = m < M ? (unsigned short)((ox * Ly + oy) * Lz + oz) : (unsigned short)0;
  }
  for (int v = t; v < L; v += NT) {
    const int hz = v % Lz, q = v / Lz;
    const int hy = q % Ly, hx = q / Ly;
    vtab[v] = hx | (hy << 8) | (hz << 16);
  }
  for (int k = t; k < a.nts * TPK; k += NT) {
    int off = 0;
    if (k < taps) {
      const int kz = k % a.KZ, q = k / a.KZ;
      const int ky = q % a.KY, kx = q / a.KY;
      off = (kx * Ly + ky) * Lz + kz;
    }
    ttab[k] = off;
  }
  __syncthreads();

  // ---- per-lane fragment geometry --------------------------------------------------------
  const int fr = lane & 15, fg = lane >> 4;
  // Activation image in LDS.  TPK = 2: voxel-major 32-byte rows [voxel][2 octets] - one DMA instruction
  // then fetches 32 B per voxel with adjacent lanes (half the L2 requests of an octet-plane gather) and
  // ds_read_b128 stays conflict-free (even/odd 16-byte slots of the two octets never meet inside a
  // 16-lane read group).  Otherwise: octet-major planes [octet][voxel][16 B].
  constexpr bool VM = TPK == 2;
  constexpr int RB = VM ? 32 : 16;  // bytes per voxel row
  const int lane_plane = VM ? (fg & 1) * 16 : (fg % PL) * a.P;
  const int lane_tsub = fg / PL;           // which of the K-step's TPK taps this lane's octet belongs to
  int hb[TM];                              // byte offset of row `fr` of m-tile i in a plane
#pragma unroll
  for (int i = 0; i < TM; ++i) hb[i] = (int)htab[(wm * TM + i) * 16 + fr] * RB + lane_plane;

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int U = a.ups ? 1 : 0;
  const int nstages = (a.nts + a.TS - 1) / a.TS;
  const int stage_units = a.TS * NTW;  // 1 KB fragments per weight stage
  const int UPP = VM ? (L + 31) >> 5 : (L + 63) >> 6;  // 1 KB DMA units per activation plane (VM: per chunk)
  const int HU = VM ? UPP : UPP * PL;                   // ... per chunk
  const int xs_bytes = VM ? a.P : PL * a.P;
  const unsigned short* wbase = a.wf + (size_t)nt0 * 512;
  typedef __attribute__((address_space(3))) char* lptr_t;
  const unsigned xs_lds = (unsigned)(unsigned long)(lptr_t)Xs;  // LDS byte addresses
  const unsigned ws_lds = (unsigned)(unsigned long)(lptr_t)Ws;

  // LDS-DMA (global_load_lds): each wave moves whole 1 KB units, lane l -> unit base + 16*l.
  // weights of stage (chunk, st) -> Ws[buf]
  auto w_issue = [&](int chunk, int st, int buf) {
    const unsigned dst = ws_lds + buf * stage_units * 1024;
    for (int u = wave; u < stage_units; u += WAVES) {
      const int tsi = u / NTW, nl = u - tsi * NTW;
      const int ts = st * a.TS + tsi;
      if (ts < a.nts && nt0 + nl < a.NT_total) {
        const unsigned short* src = wbase + ((size_t)(chunk * a.nts + ts) * a.NT_total + nl) * 512 + lane * 8;
        glds16(src, __builtin_amdgcn_readfirstlane(dst + u * 1024));
      }
    }
  };
  // The halo geometry is the same for every chunk, so each wave resolves the source of "its" DMA units
  // (u = wave + WAVES*k) once: element offset of the lane's voxel (or OOB), octet plane, LDS offset.
  constexpr int XK = 10;  // max units per wave per chunk (checked on the host)
  unsigned xoff[XK];
  int xo8[XK], xdst[XK];
#pragma unroll
  for (int k = 0; k < XK; ++k) {
    const int u = wave + WAVES * k;
    unsigned off = 0xFFFFFFFFu;
    int pl = 0, dsto = 0;
    if (u < HU) {
      int v;
      if constexpr (VM) {
        pl = lane & 1;
        v = u * 32 + (lane >> 1);
        dsto = u * 1024;
      } else {
        pl = u / UPP;
        v = (u - pl * UPP) * 64 + lane;
        dsto = pl * a.P + (u - pl * UPP) * 1024;
      }
      if (v < L) {
        const unsigned hv = vtab[v];
        const int gx = x0 - a.px + (int)(hv & 255), gy = y0 - a.py + (int)((hv >> 8) & 255),
                  gz = z0 - a.pz + (int)(hv >> 16);
        if ((unsigned)gx < (unsigned)(a.Xi << U) && (unsigned)gy < (unsigned)(a.Yi << U) &&
            (unsigned)gz < (unsigned)a.Zi) {
          const long vox = (((long)b * a.Xi + (gx >> U)) * a.Yi + (gy >> U)) * a.Zi + gz;
          off = (unsigned)(vox * a.in_ctot + a.in_off + 8 * pl);
        }
      }
    }
    xoff[k] = off;
    xo8[k] = 8 * pl;
    xdst[k] = __builtin_amdgcn_readfirstlane(dsto);
  }
  // units [u0, u1) of the activation chunk (with halo) -> Xs[buf]; out-of-range voxels read the zero page
  auto x_issue = [&](int chunk, int buf, int u0, int u1) {
    const unsigned dst = xs_lds + buf * xs_bytes;
#pragma unroll
    for (int k = 0; k < XK; ++k) {
      const int u = wave + WAVES * k;
      if (u >= u0 && u < u1) {
        const bool ok = xoff[k] != 0xFFFFFFFFu && chunk * CK + xo8[k] < a.cin_valid;
        const unsigned short* src = ok ? a.in + (size_t)xoff[k] + chunk * CK
                                       : reinterpret_cast<const unsigned short*>(a.zero16);
        glds16(src, dst + xdst[k]);
      }
    }
  };

  x_issue(0, 0, 0, HU);
  w_issue(0, 0, 0);
  dma_wait();
  __syncthreads();

  const int total_phases = a.nchunks * nstages;
  int chunk = 0, st = 0;
  for (int ph = 0; ph < total_phases; ++ph) {
    // ---- prefetch: next weight stage, and a slice of the next chunk's activations.  The burst costs each
    // wave several hundred issue cycles during which it feeds no MFMAs, so the two halves of the workgroup
    // (waves w and w + WAVES/2 share a SIMD) take turns: the first half issues at the top of the phase, the
    // second half in the middle of its K-step loop, and the other wave keeps the SIMD's matrix pipe busy.
    if (a.xbufs != 2 && st == 0 && chunk > 0) {  // single activation buffer: reload it between chunks
      x_issue(chunk, 0, 0, HU);
      dma_wait();
      __syncthreads();
    }
    auto burst = [&]() {
      if (ph + 1 < total_phases) {
        const bool wrap = st + 1 == nstages;
        w_issue(wrap ? chunk + 1 : chunk, wrap ? 0 : st + 1, (ph + 1) & 1);
      }
      if (a.xbufs == 2 && chunk + 1 < a.nchunks)
        x_issue(chunk + 1, (chunk + 1) & 1, (HU * st) / nstages, (HU * (st + 1)) / nstages);
    };

    const char* wcur = Ws + (ph & 1) * stage_units * 1024 + (wn * TN) * 1024 + lane * 16;
    const char* xcur = Xs + (a.xbufs == 2 ? (chunk & 1) * xs_bytes : 0);
    const int ts_end = min(a.TS, a.nts - st * a.TS);
    const int* tt = ttab + st * a.TS * TPK + lane_tsub;
    auto load_frags = [&](int tsi, uint4 (&wf)[TN], uint4 (&xf)[TM]) {
      const int toff = tt[tsi * TPK] * RB;
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const uint4*>(wcur + (tsi * NTW + j) * 1024);
#pragma unroll
      for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const uint4*>(xcur + hb[i] + toff);
    };
    auto mma_frags = [&](const uint4 (&wf)[TN], const uint4 (&xf)[TM]) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_chunk<BF16>(acc[i][j], wf[j], xf[i]);
    };
    auto run_ksteps = [&](int lo, int hi) {
      if (lo >= hi) return;
      if constexpr (PIPE) {
        // register double-buffering: the fragments of K-step t+1 are in flight during the MFMAs of K-step t
        uint4 wA[TN], xA[TM], wB[TN], xB[TM];
        load_frags(lo, wA, xA);
        int tsi = lo;
        for (; tsi + 2 <= hi; tsi += 2) {
          load_frags(tsi + 1, wB, xB);
          mma_frags(wA, xA);
          if (tsi + 2 < hi) load_frags(tsi + 2, wA, xA);
          mma_frags(wB, xB);
        }
        if (tsi < hi) mma_frags(wA, xA);
      } else {
        for (int tsi = lo; tsi < hi; ++tsi) {
          uint4 wf[TN], xf[TM];
          load_frags(tsi, wf, xf);
          mma_frags(wf, xf);
        }
      }
    };
    const int mid = (STAGGER && wave >= WAVES / 2) ? ts_end / 2 : 0;  // wave-uniform
    if (mid == 0) burst();
    run_ksteps(0, mid);
    if (mid > 0) burst();
    run_ksteps(mid, ts_end);
    dma_wait();  // this wave's prefetches have landed ...
    __syncthreads();  // ... and so have everybody else's; the buffers just read are free again
    if (++st == nstages) { st = 0; ++chunk; }
  }

  // ---- epilogue: acc[i][j][r] -> channel (nt0 + wn*TN + j)*16 + 4*fg + r, voxel row fr of m-tile i
  const long vox_per_b = (long)a.Xo * a.Yo * a.Zo;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const unsigned mv = mtab[(wm * TM + i) * 16 + fr];
    const int gx = x0 + (int)(mv & 255), gy = y0 + (int)((mv >> 8) & 255), gz = z0 + (int)((mv >> 16) & 255);
    if ((mv >> 24) || gx >= a.Xo || gy >= a.Yo || gz >= a.Zo) continue;
    const long vi = ((long)gx * a.Yo + gy) * a.Zo + gz;
    const long m = (long)b * vox_per_b + vi;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int co0 = (nt0 + wn * TN + j) * 16 + fg * 4;
      if (co0 >= a.Cout) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      const int nval = (a.Cout - co0) < 4 ? (a.Cout - co0) : 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < nval) {
          float x = v[q];
          if (a.bias) x += a.bias[co0 + q];
          if (a.act) x = x > 0.f ? x : x * a.slope;
          if (a.chan_scale) x *= a.chan_scale[(long)b * a.Cout + co0 + q];
          v[q] = x * a.alpha;
        }
      }
      if (a.out_planar) {
        float* o = reinterpret_cast<float*>(a.out);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < nval) o[((long)b * a.Cout + co0 + q) * vox_per_b + vi] = v[q];
      } else {
        unsigned short* o = reinterpret_cast<unsigned short*>(a.out) + m * a.out_ctot + a.out_off + co0;
        const unsigned short* rp = a.res ? a.res + m * a.res_ctot + a.res_off + co0 : nullptr;
        // LeakyReLU backward of the layer whose output gradient this is (channels [mask_c0, mask_c1)): the
        // multiply by (y > 0 ? 1 : slope) happens here, after the accumulation, instead of in its own pass
        const unsigned short* mp = nullptr;  // (compiled in only for MASK: it costs ~25 VGPRs in the epilogue)
        if constexpr (MASK) {
          if (co0 >= a.mask_c0 && co0 < a.mask_c1) mp = a.mask_y + m * a.mask_ctot + a.mask_off + (co0 - a.mask_c0);
        }
        if (nval == 4 && a.vec_ok) {
          float4 o4 = make_float4(v[0], v[1], v[2], v[3]);
          if (rp) {
            const float4 r4 = ld4<BF16>(rp);
            o4.x += a.beta * r4.x;
            o4.y += a.beta * r4.y;
            o4.z += a.beta * r4.z;
            o4.w += a.beta * r4.w;
          }
          if (mp) {
            const float4 y4 = ld4<BF16>(mp);
            o4.x *= y4.x > 0.f ? 1.f : a.mask_slope;
            o4.y *= y4.y > 0.f ? 1.f : a.mask_slope;
            o4.z *= y4.z > 0.f ? 1.f : a.mask_slope;
            o4.w *= y4.w > 0.f ? 1.f : a.mask_slope;
          }
          st4<BF16>(o, o4);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q < nval) {
              float x = v[q];
              if (rp) x += a.beta * ldf<BF16>(rp + q);
              if (mp && co0 + q < a.mask_c1) x *= ldf<BF16>(mp + q) > 0.f ? 1.f : a.mask_slope;
              stf<BF16>(o + q, x);
            }
        }
      }
    }
  }
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

template <int WM, int WN, int TM, int TN, int TPK, bool MASK = false>
int launch_ct(CtArgs& a, hipStream_t st) {
  constexpr int WAVES = WM * WN, NTW = WN * TN;
  const int taps = a.KX * a.KY * a.KZ;
  constexpr int M = WM * TM * 16;  // table sizes follow the MFMA rows; the tile volume may be smaller
  if (a.TX * a.TY * a.TZ > M) return WSR_EUNSUPPORTED;
  const int L = (a.TX + a.KX - 1) * (a.TY + a.KY - 1) * (a.TZ + a.KZ - 1);
  if (L > 65535) return WSR_EUNSUPPORTED;
  a.nts = (taps + TPK - 1) / TPK;
  a.NT_total = (a.Cout + 15) / 16;
  a.ngroups = (a.NT_total + NTW - 1) / NTW;
  constexpr int PL = 4 / TPK;
  constexpr bool VM = TPK == 2;
  a.P = VM ? round_up(L * 32, 1024) : round_up(L * 16, 1024);  // whole 1 KB DMA units; == 0 (mod 256)
  a.off_mtab = 0;
  a.off_htab = M * 4;
  a.off_vtab = round_up(a.off_htab + M * 2, 16);
  a.off_ttab = a.off_vtab + L * 4;
  a.off_xs = round_up(a.off_ttab + a.nts * TPK * 4, 1024);
  // two activation buffers when that still leaves room for weight stages of >= 2 K-steps
  int ts_max = 0;
  for (a.xbufs = 2; a.xbufs >= 1; --a.xbufs) {
    a.off_ws = a.off_xs + a.xbufs * (VM ? 1 : PL) * a.P;
    const int avail = 160 * 1024 - a.off_ws;
    ts_max = avail / (2 * NTW * 1024);
    const int cap_kb = getenv("WSR_WSTAGE_KB") ? atoi(getenv("WSR_WSTAGE_KB")) : 48;  // tuning aid
    const int cap = cap_kb / NTW > 0 ? cap_kb / NTW : 1;  // <= 48 KB per weight stage (measured: up-convs +7 %, others flat)
    if (ts_max > cap) ts_max = cap;
    if (ts_max >= 2 || (ts_max >= 1 && a.nts == 1)) break;
  }
  if (a.xbufs < 1) {
    a.xbufs = 1;
    if (ts_max < 1) return WSR_EUNSUPPORTED;
  }
  if ((VM ? (L + 31) / 32 : ((L + 63) / 64) * PL) > 10 * WAVES) return WSR_EUNSUPPORTED;  // XK units per wave
  if ((long)a.B * a.Xi * a.Yi * a.Zi * a.in_ctot >= 0xFFFFFFFFL) return WSR_EUNSUPPORTED;  // 32-bit element offsets
  const int nph = (a.nts + ts_max - 1) / ts_max;
  a.TS = (a.nts + nph - 1) / nph;  // balanced stages
  const size_t lds = (size_t)a.off_ws + (size_t)2 * a.TS * NTW * 1024;
  if (lds > 160 * 1024) return WSR_EUNSUPPORTED;
  a.tiles_x = (a.Xo + a.TX - 1) / a.TX;
  a.tiles_y = (a.Yo + a.TY - 1) / a.TY;
  a.tiles_z = (a.Zo + a.TZ - 1) / a.TZ;
  a.ntiles = a.B * a.tiles_x * a.tiles_y * a.tiles_z;
  constexpr bool PIPE = TN <= 7;  // register budget: (TM+TN)*8 fragment + TM*TN*4 accumulator VGPRs
  if (a.mask_y && !MASK) return WSR_EUNSUPPORTED;
  auto kern = conv_tile_kernel<WM, WN, TM, TN, TPK, PIPE, MASK>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.ntiles * a.ngroups)), dim3(WAVES * 64), lds, st, a);
  WSR_LAUNCH_CHECK();
  return 0;
}

// choose the spatial tile: at most M voxels, z (the contiguous axis) kept whole when it is short
static void pick_tile(CtArgs& a, int M) {
  int tz;
  if (a.Zo <= 16) tz = a.Zo;
  else if (a.Zo % 16 == 0 || a.Zo > 64) tz = 16;
  else tz = 8;
  while (tz > M) tz >>= 1;
  const int rest = M / tz;
  int tx = 1, ty = 1;
  if ((rest & (rest - 1)) == 0) {
    while (tx * ty < rest) {  // power of two: balanced split, y first
      if (ty <= tx) ty <<= 1; else tx <<= 1;
    }
  } else {
    while ((tx + 1) * (tx + 1) <= rest) ++tx;
    ty = rest / tx;
  }
  a.TX = tx; a.TY = ty; a.TZ = tz;
}


}  // namespace
