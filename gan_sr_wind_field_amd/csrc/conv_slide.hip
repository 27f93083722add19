// Memory-bound z-tapless convolutions with a thin side, bf16: a workgroup SLIDES along x.
//
// The last conv of the generator (reference Generator_3D_Resnet_ESRGAN.py:105-110, hr_convs[2]: 5x5x5, 144 -> 3)
// runs z-folded as a (5,5,1) conv with 15 outputs.  Both of its passes move a 144-channel HR tensor
// once and do little arithmetic per byte; on the generic halo-tile kernel they were bound by per-tile overheads -
// the forward re-fetched the halo image and the whole 117 KB filter for every 512-voxel tile, in nine dependent
// DMA phases with one workgroup per CU (524 us for 730 MB), the input gradient spent its time in 8-byte strided
// mask loads and stores (793 us for 1.28 GB).
//
//   forward  (wsr_conv_slide_fwd):  y[v, n] = sum_{kx,ky,c} x[v + (kx,ky) - pad, c] * w[n, (kx,ky), c],  N <= 16
//
// A workgroup owns a column of the volume - TY = 16 rows x TZ = 4 levels, every x of its segment - and streams the
// x-planes of the input (all channels, rows with their y halo) through four LDS buffers: every input byte is
// fetched ONCE per column (y halo: 20/16), as whole 1152-byte runs (4 z-contiguous voxels x 288 B), and every
// fragment is read from LDS once and used for the KX output planes it contributes to (input-stationary).  The
// filter never touches LDS: the reduction (taps x channel octets) is split four ways over the waves and every wave
// keeps the fragments of its K-steps in REGISTERS for the whole launch (KX * 6 * 4 = 120 VGPRs at 144 channels).
// The four partial sums of an m-tile meet in LDS and are added in a fixed order (bit-reproducible).  Output:
// planar fp32 partial sums (the z-fold pass adds the bias).
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

__device__ __forceinline__ void cs_glds16(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_addr)
      : "memory");
}
__device__ __forceinline__ void cs_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// LDS-DMA through a buffer descriptor: LDS[lds_addr + 16*lane] <- 16 bytes at srd.base + voff, ZEROS where voff is
// outside [0, srd.num_records) - the halo rows and planes outside the tensor cost no address arithmetic at all (a
// global_load_lds needs a 64-bit per-lane address and a zero-page select: ~8 vector instructions per DMA instruction,
// and a DMA wave issues six per plane).  The descriptor words come from scalar arithmetic: s_nop 4 covers the
// SALU-write -> VMEM-read hazard inside the statement (hipcc pads nothing inside asm).
typedef __attribute__((ext_vector_type(4))) int cs_srd_t;
__device__ __forceinline__ void cs_bufdma16(cs_srd_t srd, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %2, %1, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(srd), "v"(voff), "s"(lds_addr)
      : "memory");
}
__device__ __forceinline__ cs_srd_t cs_make_srd(const void* base, unsigned bytes) {
  const unsigned long long b = (unsigned long long)base;
  cs_srd_t r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r.y = __builtin_amdgcn_readfirstlane((int)((b >> 32) & 0xFFFFu));  // stride 0
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
constexpr unsigned CS_OOB = 0x7FFFFFF0u;  // a byte offset no plane reaches (the launch checks): reads as zeros

struct CsArgs {
  const unsigned short* in;  // NDHWC bf16
  const unsigned short* wf;  // tile-kernel fragment filter, TPK = 2, one n-tile: [chunk16][tap pair][lane][8]
  float* out;                // planar fp32 (B, N, X, Y, Z)
  const void* zero16;
  const float* bias;         // [N] or NULL
  int B, X, Y, Z;
  int in_ctot, in_off;
  int N;
  int px, py;
  int nty, ntz, nseg, XS;    // tiles along y and z, segments along x and their length
  int blk;                   // 1: workgroup ids are dealt in blocks of 4 (y) x 8 (z) tiles (one XCD's share)
  unsigned long long* stamps;  // -DWSR_CS_STAMPS builds: [workgroup][wave 8][8] cycle sums per loop phase (else unused)
};

#if defined(WSR_CS_STAMPS) && WSR_CS_STAMPS != 2
#define CS_T(var) const long long var = clock64()
#define CS_ACC(k, t0, t1) st_sum[k] += (t1) - (t0)
#else
#define CS_T(var) do {} while (0)
#define CS_ACC(k, t0, t1) do {} while (0)
#endif

constexpr int cs_round_up(int v, int m) { return (v + m - 1) / m * m; }

template <int KX, int KY, int OC>
struct CsGeom {
  static constexpr int TY = 16, TZ = 4;
  static constexpr int PY = TY + KY - 1;          // rows of an input plane (with y halo)
  static constexpr int VOX = PY * TZ;             // voxels of an input plane
  static constexpr int ROWB = OC * 16;            // bytes per voxel
  static constexpr int NP = VOX * OC;             // 16-byte pieces per plane
  static constexpr int NU = (NP + 63) / 64;       // DMA units (64 pieces) per plane; the last one is shifted back
  static constexpr int PLANE_B = NP * 16;
  static constexpr int NB = 6;                    // plane buffers: one being contracted, two landed, three in flight
  static constexpr int PAIRS = KY * OC;           // (ky, octet) pairs per kx
  static constexpr int SK = (PAIRS + 3) / 4;      // K-steps per kx
  static constexpr int NPW = (SK + 3) / 4;        // K-steps per kx and wave (four K groups)
  static constexpr int NTS = (KX * KY + 1) / 2;   // tap pairs of the packed filter (TPK = 2)
  static constexpr int SCR_B = 12 * 1024;         // partial sums of one plane: 12 (K group, m-tile) pairs x 1 KB
  static constexpr int LDS_B = NB * PLANE_B + 2 * SCR_B;
};

template <int KX, int KY, int OC>
__global__ __launch_bounds__(512) void conv_slide_fwd_kernel(const CsArgs a) {
  using G = CsGeom<KX, KY, OC>;
  constexpr int TY = G::TY, TZ = G::TZ, ROWB = G::ROWB, NP = G::NP, NU = G::NU, PLANE_B = G::PLANE_B;
  constexpr int PAIRS = G::PAIRS, SK = G::SK, NPW = G::NPW, NTS = G::NTS, SCR_B = G::SCR_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int kg = wave & 3, mh = wave >> 2;  // K group, m-tile half (m-tiles 2mh, 2mh + 1)
  const int fr = lane & 15, fg = lane >> 4;
#ifdef WSR_CS_STAMPS
  const long long st_wall0 = wall_clock64();  // (100 MHz)
#endif

  // ---- which column --------------------------------------------------------------------------------
  unsigned bid = (unsigned)xcd_remap(blockIdx.x, gridDim.x);
  int ty, tz;
  if (a.blk) {  // blocks of 4 x 8 tiles: y neighbours (shared halo rows) and z neighbours (shared output lines)
    const unsigned within = bid & 31u;
    unsigned r = bid >> 5;
    const unsigned nzb = (unsigned)a.ntz >> 3, nyb = (unsigned)a.nty >> 2;
    const unsigned zb = r % nzb; r /= nzb;
    const unsigned yb = r % nyb; r /= nyb;
    ty = (int)(yb * 4 + (within >> 3));
    tz = (int)(zb * 8 + (within & 7u));
    bid = r;
  } else {
    tz = (int)(bid % (unsigned)a.ntz); bid /= (unsigned)a.ntz;
    ty = (int)(bid % (unsigned)a.nty); bid /= (unsigned)a.nty;
  }
  const int seg = (int)(bid % (unsigned)a.nseg);
  const int b = (int)(bid / (unsigned)a.nseg);
  const int y0 = ty * TY, z0 = tz * TZ;
  const int x_begin = seg * a.XS;
  const int nplanes = min(a.XS, a.X - x_begin);  // output planes of this workgroup (>= 1 by construction)

  typedef __attribute__((address_space(3))) char* lptr_t;
  const unsigned ring_lds = (unsigned)(unsigned long)(lptr_t)smem;
  char* scratch = smem + G::NB * PLANE_B;

  // ---- DMA geometry: the source of this lane's piece of each of the wave's units (same for every plane) ---------
  // The planes are fetched by the waves of K groups 2 and 3; the waves of K groups 0 and 1 finalize and store the
  // outputs.  Stores and LDS-DMA share the in-order vmcnt counter: a wave that did both waited, every plane, for the
  // previous plane's 16-byte output stores to be acknowledged by memory before it could tell that its DMA had
  // landed - 2.6 us per plane, three times the arithmetic.  Now no wave ever waits for a store.
  const bool dma_wave = kg >= 2;
  const int dw = (kg - 2) + 2 * mh;  // 0 .. 3 among the DMA waves
  constexpr int UPW = (NU + 3) / 4;
  unsigned uoff[UPW];  // byte offset inside an x-plane of the input, or CS_OOB (reads as zeros)
  unsigned udst[UPW];  // LDS byte offset of the unit inside a plane (wave-uniform)
#pragma unroll
  for (int k = 0; k < UPW; ++k) {
    const int u = dw + 4 * k;
    unsigned off = CS_OOB;
    unsigned dst = 0;
    if (dma_wave && u < NU) {
      const int p0 = u * 64 < NP - 64 ? u * 64 : NP - 64;  // the last unit is shifted back: no partial unit
      dst = (unsigned)p0 * 16u;
      const int p = p0 + lane;
      const int vox = p / OC, o = p - vox * OC;
      const int yl = vox / TZ, zl = vox - yl * TZ;
      const int gy = y0 - a.py + yl, gz = z0 + zl;
      if ((unsigned)gy < (unsigned)a.Y && gz < a.Z) off = (unsigned)((gy * a.Z + gz) * a.in_ctot + a.in_off + o * 8) * 2u;
    }
    uoff[k] = off;
    udst[k] = __builtin_amdgcn_readfirstlane(dst);
  }
  const long plane_elems = (long)a.Y * a.Z * a.in_ctot;
  // descriptor of input plane xi: a plane outside the tensor (or past the segment's last one) has zero records -
  // every lane reads zeros
  auto plane_srd = [&](int xi, bool wanted) __attribute__((always_inline)) {
    const bool xin = wanted && (unsigned)xi < (unsigned)a.X;
    return cs_make_srd(a.in + ((long)b * a.X + (xin ? xi : 0)) * plane_elems, xin ? (unsigned)(plane_elems * 2) : 0u);
  };
  auto plane_issue = [&](int xi, int slot, bool wanted) __attribute__((always_inline)) {
    if (!dma_wave) return;
    const cs_srd_t srd = plane_srd(xi, wanted);
    const unsigned dst = ring_lds + (unsigned)slot * PLANE_B;
#pragma unroll
    for (int k = 0; k < UPW; ++k)
      if (dw + 4 * k < NU) cs_bufdma16(srd, uoff[k], dst + udst[k]);
  };

  // ---- this wave's filter fragments, in registers for the whole launch --------------------------------
  // K-step sl (0 .. SK-1) of tap column kx contracts the four (ky, octet) pairs q = 4 sl + fg, one per lane group;
  // consecutive pairs alternate the octet's parity, which keeps the 288-byte voxel rows conflict-free for
  // ds_read_b128.  The wave takes the K-steps sl = 4 j + kg.
  uint4 wreg[KX][NPW];
  int woff[NPW];
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    const int sl = 4 * j + kg;
    const int q = 4 * sl + fg;
    const bool ok = sl < SK && q < PAIRS;
    const int ky = ok ? q / OC : 0, o = ok ? q - (q / OC) * OC : 0;
    // byte offset of this lane's fragment row inside a plane: voxel (y = 4 m + fr/4 + ky, z = fr%4) = m*16 + fr + ky*TZ
    woff[j] = (2 * mh * 16 + fr + ky * TZ) * ROWB + o * 16;
#pragma unroll
    for (int kx = 0; kx < KX; ++kx) {
      uint4 w = make_uint4(0u, 0u, 0u, 0u);
      if (ok) {
        const int tap = kx * KY + ky;
        const int g = ((tap & 1) << 1) | (o & 1);
        w = *reinterpret_cast<const uint4*>(a.wf + ((size_t)((o >> 1) * NTS + (tap >> 1)) * 64 + g * 16 + fr) * 8);
      }
      wreg[kx][j] = w;
    }
  }
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * fg + r < a.N) bias4[r] = a.bias[4 * fg + r];
  }

  // ---- schedule: input-stationary ----------------------------------------------------------------------------
  // Input plane p feeds output planes p - kx (tap column kx).  A fragment of plane p is read from LDS ONCE and
  // multiplied into the accumulators of all KX outputs it belongs to; an output's accumulator is complete KX planes
  // after it was opened and the set shifts by one.  (Output-stationary - one x fragment per MFMA, every plane read
  // KX times from a KX + 1 deep ring - saturated the LDS read path and the matrix pipe at the same time and got the
  // sum of the two, not the maximum: 340 us with both ~45 % busy.)  LDS holds NB = 6 planes: the one being contracted,
  // the next one (landed and published one barrier early, so that its first fragments are requested BEFORE the
  // barrier that ends the iteration and the matrix work restarts without an LDS round trip) and three in flight
  // (~69 KB per CU on the wire).
  // The loop body is straight-line: every iteration issues a plane (past the end: a zero-record descriptor, no
  // traffic), writes / sums partial sums (before the first complete output: of garbage, never stored) - the first
  // version chose among these per iteration with ~150 scalar branches and spent 0.6 us per plane on them.
  constexpr int NB = G::NB;
  const int n_in = nplanes + KX - 1;   // input planes the workgroup contracts
  const int xin0 = x_begin - a.px;     // global x of input plane 0
  const int nw = dma_wave ? (NU - dw + 3) / 4 : 0;  // DMA instructions this wave issues per plane (wave-uniform)
  constexpr int NWMAX = (NU + 3) / 4;
  static_assert(NWMAX >= 2 && (NB - 3) * NWMAX <= 63, "counted DMA wait out of range");
  // wait until everything but the `young` most recently issued planes of this wave has landed (nw or nw - 1 DMA
  // instructions per plane, by wave)
  auto wait_landed = [&](auto young_c) __attribute__((always_inline)) {
    constexpr int YOUNG = decltype(young_c)::value;
    if (!dma_wave) return;  // (its vmcnt holds output stores only: nothing to wait for)
    if (nw == NWMAX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNG * NWMAX) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNG * (NWMAX - 1)) : "memory");
  };
  using young_loop = std::integral_constant<int, NB - 3>;

  // ---- prologue: planes 0 .. NB-2 are issued, planes 0 and 1 land ---------------------------------------------------
#pragma unroll
  for (int q = 0; q < NB - 1; ++q) plane_issue(xin0 + q, q, q < n_in);
  wait_landed(young_loop{});
  __syncthreads();

  // finalizer of m-tile m: wave (kg = m & 1, mh = m >> 1); everybody else stores its partial sums.  The sum of
  // output plane o is taken one iteration LATER, under the MFMAs of the next input plane: its LDS reads are requested
  // right after the barrier that published the partial sums and consumed behind the matrix work.
  const int my_m0 = 2 * mh, my_m1 = 2 * mh + 1;
  const bool fin0 = kg == (my_m0 & 1), fin1 = kg == (my_m1 & 1), fin = fin0 || fin1;
  const int fin_m = fin0 ? my_m0 : my_m1, fin_k = fin_m & 1;
  // partial-sum slots: [m-tile 4][K group minus the finalizer's: 3][lane][16 B]; finalizer K group of m-tile m is m & 1
  auto pslot = [&](int m, int g) __attribute__((always_inline)) {
    const int fk = m & 1;
    return (m * 3 + (g > fk ? g - 1 : g)) * 1024 + lane * 16;
  };
  const long out_plane = (long)a.Y * a.Z;  // voxels of one (b, n, x) plane
  const long nstride = (long)a.X * out_plane;
  const int fgy = y0 + 4 * fin_m + (fr >> 2), fgz = z0 + (fr & 3);
  const bool fin_vox = fin && fgy < a.Y && fgz < a.Z;
  float* const out_lane = a.out + ((long)b * a.N + 4 * fg) * nstride + (long)fgy * a.Z + fgz;
  // byte offsets of the partial sums this wave reads as finalizer (K groups other than its own), and writes otherwise
  int rd_off[3];
#pragma unroll
  for (int g3 = 0; g3 < 3; ++g3) rd_off[g3] = (fin_m * 3 + g3) * 1024 + lane * 16;
  const int wr0 = pslot(my_m0, kg), wr1 = pslot(my_m1, kg);
  auto finalize = [&](const f32x4_t (&part)[3], const f32x4_t& own, int o) __attribute__((always_inline)) {
    // fixed order over the K groups: part[] holds groups {0..3} \ {fin_k} in order, own sits at position fin_k
    f32x4_t sum;
    if (fin_k == 0) {
      sum = own;
#pragma unroll
      for (int g3 = 0; g3 < 3; ++g3) { sum[0] += part[g3][0]; sum[1] += part[g3][1]; sum[2] += part[g3][2]; sum[3] += part[g3][3]; }
    } else {  // fin_k == 1
      sum = part[0];
      sum[0] += own[0]; sum[1] += own[1]; sum[2] += own[2]; sum[3] += own[3];
#pragma unroll
      for (int g3 = 1; g3 < 3; ++g3) { sum[0] += part[g3][0]; sum[1] += part[g3][1]; sum[2] += part[g3][2]; sum[3] += part[g3][3]; }
    }
    if (fin_vox && o >= 0) {
      float* op = out_lane + (long)(x_begin + o) * out_plane;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * fg + r < a.N) op[r * nstride] = sum[r] + bias4[r];
    }
  };

  f32x4_t acc[KX][2];                      // acc[kx]: output plane p - kx, m-tiles 2 mh and 2 mh + 1
#pragma unroll
  for (int kx = 0; kx < KX; ++kx) acc[kx][0] = acc[kx][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  f32x4_t own_prev = {0.f, 0.f, 0.f, 0.f};  // the finalizer's own partial sum of the plane completed last iteration
  int slot = 0;  // p mod NB
  // x fragments: a ring of D steps (the whole plane's 2 NPW fragments at once would not fit beside the filter); the
  // first D of a plane are requested at the end of the previous iteration
  constexpr int D = 3;
  uint4 xa[D], xb[D];
#pragma unroll
  for (int j = 0; j < D && j < NPW; ++j) {
    xa[j] = *reinterpret_cast<const uint4*>(smem + woff[j]);
    xb[j] = *reinterpret_cast<const uint4*>(smem + woff[j] + 16 * ROWB);
  }
#ifdef WSR_CS_STAMPS
  long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long st_wall1 = wall_clock64();
  const long long st_begin = clock64();
#endif
  // The iteration's one barrier sits in the MIDDLE of the plane's matrix work (after MB of its NPW MFMA groups): a wave
  // that arrives early waits while its SIMD partner still issues MFMAs, and the serial part of a plane - partial sums
  // to LDS, first fragments of the next plane, the wait for the landed DMA - runs under the partner's MFMAs instead of
  // with an empty pipe (at the end of the plane it cost ~550 of 2 750 cycles per plane).  What the barrier orders:
  //  * partial sums written at the end of iteration p-1  ->  summed by the finalizers after the barrier of iteration p;
  //  * every wave is done with plane p-1  ->  its buffer takes plane p + NB - 1 (DMA issued behind the later groups);
  //  * plane p+1 has landed (counted wait of the DMA waves in front of the barrier)  ->  its first fragments are
  //    requested at the end of iteration p.
  constexpr int MB = 2;
  static_assert(NPW > MB + 1, "the groups behind the barrier carry the DMA issue and the finalizer");
  constexpr int UPG = (UPW + (NPW - MB) - 1) / (NPW - MB);  // DMA instructions behind each MFMA group after the barrier
  for (int p = 0; p < n_in; ++p) {
    CS_T(t0);
    int s_nxt = slot + NB - 1;
    if (s_nxt >= NB) s_nxt -= NB;
    const cs_srd_t srd_nxt = plane_srd(xin0 + p + NB - 1, p + NB - 1 < n_in);
    const unsigned dst_nxt = ring_lds + (unsigned)s_nxt * PLANE_B;
    CS_T(t1);
    CS_ACC(0, t0, t1);
    const char* pl = smem + slot * PLANE_B;
    f32x4_t part[3];
    const char* scr_prev = scratch + ((p - 1) & 1) * SCR_B;
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      if (j == MB) {
        wait_landed(young_loop{});  // plane p + 1 has landed; the three planes issued after it stay in flight
        __syncthreads();
        // partial sums of output plane p - KX (completed by input plane p - 1): requested now, summed behind the MFMAs
        if (fin) {
#pragma unroll
          for (int g3 = 0; g3 < 3; ++g3) part[g3] = *reinterpret_cast<const f32x4_t*>(scr_prev + rd_off[g3]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef WSR_CS_ABL_NOMFMA
      if (a.N > 1000)  // (tuning build: the matrix work is skipped, everything else runs)
#endif
      if (j == 0) {
        // The accumulator set SHIFTS here, for free: the first MFMA of output plane p - kx takes the sums of tap
        // columns < kx from the registers of acc[kx - 1] as its C operand and writes acc[kx] (descending kx: every
        // source is read before it is overwritten); acc[0] opens from zero.
#pragma unroll
        for (int kx = KX - 1; kx >= 0; --kx) {
          const f32x4_t c0 = kx ? acc[kx - 1][0] : f32x4_t{0.f, 0.f, 0.f, 0.f};
          const f32x4_t c1 = kx ? acc[kx - 1][1] : f32x4_t{0.f, 0.f, 0.f, 0.f};
          acc[kx][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wreg[kx][0]),
                                                               __builtin_bit_cast(bf16x8_t, xa[0]), c0, 0, 0, 0);
          acc[kx][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wreg[kx][0]),
                                                               __builtin_bit_cast(bf16x8_t, xb[0]), c1, 0, 0, 0);
        }
      } else {
        // (descending kx in every group: the accumulators that complete with this plane, acc[KX - 1], get their last
        // MFMAs first and are out of the pipe when the partial sums are written below)
#pragma unroll
        for (int kx = KX - 1; kx >= 0; --kx) {
          mma_chunk<BF16>(acc[kx][0], wreg[kx][j], xa[j % D]);
          mma_chunk<BF16>(acc[kx][1], wreg[kx][j], xb[j % D]);
        }
      }
      if (j + D < NPW) {
        xa[j % D] = *reinterpret_cast<const uint4*>(pl + woff[j + D]);
        xb[j % D] = *reinterpret_cast<const uint4*>(pl + woff[j + D] + 16 * ROWB);
      }
      __builtin_amdgcn_sched_barrier(0);
      // The DMA instructions of plane p + NB - 1 are issued BETWEEN the MFMA groups, UPG per group: issued in a block
      // they cost the DMA waves 600-800 cycles per plane in front of their matrix work (in-kernel stamps), and the
      // output stores of the finalizing waves queued behind them for another ~1 300.
#ifdef WSR_CS_ABL_NODMA
      if (a.N > 1000)  // (tuning build: no plane is fetched after the prologue)
#endif
      if (j >= MB) {
#pragma unroll
        for (int k = (j - MB) * UPG; k < (j - MB + 1) * UPG && k < UPW; ++k)
          if (dma_wave && dw + 4 * k < NU) cs_bufdma16(srd_nxt, uoff[k], dst_nxt + udst[k]);
      }
      if (j == MB + 1 && fin) finalize(part, own_prev, p - KX);  // (its LDS reads have long landed)
    }
    __builtin_amdgcn_sched_barrier(0);
    CS_T(t2);
    CS_ACC(1, t1, t2);
    // ---- output plane p - (KX - 1) is complete: partial sums -> LDS (the set shifts at the first MFMAs of the next plane)
    const f32x4_t acc0 = acc[KX - 1][0], acc1 = acc[KX - 1][1];
    char* scr = scratch + (p & 1) * SCR_B;
    if (!fin0) *reinterpret_cast<f32x4_t*>(scr + wr0) = acc0;
    if (!fin1) *reinterpret_cast<f32x4_t*>(scr + wr1) = acc1;
    own_prev = fin0 ? acc0 : acc1;
    {  // first fragments of plane p + 1 (published by this iteration's barrier)
      int sn = slot + 1;
      if (sn >= NB) sn -= NB;
      const char* pn = smem + sn * PLANE_B;
#pragma unroll
      for (int j = 0; j < D && j < NPW; ++j) {
        xa[j] = *reinterpret_cast<const uint4*>(pn + woff[j]);
        xb[j] = *reinterpret_cast<const uint4*>(pn + woff[j] + 16 * ROWB);
      }
    }
    CS_T(t4);
    CS_ACC(3, t2, t4);
    if (++slot == NB) slot = 0;
  }
  __syncthreads();  // the last plane's partial sums
#ifdef WSR_CS_STAMPS
  if (a.stamps && lane == 0) {
    unsigned long long* q = a.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
#if WSR_CS_STAMPS == 2  // wall-clock mode: where the launch's time goes around the loop
    q[0] = (unsigned long long)st_wall0;
    q[1] = (unsigned long long)st_wall1;
    q[2] = (unsigned long long)wall_clock64();
#else
    for (int k = 0; k < 6; ++k) q[k] = (unsigned long long)st_sum[k];
#endif
    q[6] = (unsigned long long)(clock64() - st_begin);
    q[7] = (unsigned long long)n_in;
  }
#endif
  if (fin) {  // the last output plane
    f32x4_t part[3];
    const char* scr_prev = scratch + ((n_in - 1) & 1) * SCR_B;
#pragma unroll
    for (int g3 = 0; g3 < 3; ++g3) part[g3] = *reinterpret_cast<const f32x4_t*>(scr_prev + rd_off[g3]);
    finalize(part, own_prev, n_in - KX);
  }
  cs_dma_wait();  // (zero-record planes issued past the end still write LDS: they must not outlive the workgroup)
#if defined(WSR_CS_STAMPS) && WSR_CS_STAMPS == 2
  if (a.stamps && lane == 0) a.stamps[((size_t)blockIdx.x * 8 + wave) * 8 + 3] = (unsigned long long)wall_clock64();
#endif
}

template <int KX, int KY, int OC>
int launch_slide_fwd(CsArgs& a, hipStream_t st) {
  using G = CsGeom<KX, KY, OC>;
  static_assert(G::LDS_B <= 160 * 1024, "ring + scratch exceed the LDS");
  static_assert(G::NP >= 64, "plane smaller than one DMA unit");
  auto kern = conv_slide_fwd_kernel<KX, KY, OC>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       G::LDS_B);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  a.nty = (a.Y + G::TY - 1) / G::TY;
  a.ntz = (a.Z + G::TZ - 1) / G::TZ;
  const long cols = (long)a.B * a.nty * a.ntz;
  // segments along x: enough workgroups to fill the chip, each at least 8 planes long (KX - 1 halo planes each)
  int nseg = (int)((256 + cols - 1) / cols);
  if (nseg > a.X / 8) nseg = a.X / 8;
  if (nseg < 1) nseg = 1;
  a.XS = (a.X + nseg - 1) / nseg;
  a.nseg = (a.X + a.XS - 1) / a.XS;
  a.blk = (a.nty % 4 == 0 && a.ntz % 8 == 0) ? 1 : 0;
  const long wg = cols * a.nseg;
  if (wg >= (1l << 31)) return WSR_EUNSUPPORTED;
#ifdef WSR_CS_STAMPS
  a.stamps = getenv("WSR_CS_STAMPS_PTR") ? (unsigned long long*)strtoull(getenv("WSR_CS_STAMPS_PTR"), nullptr, 0) : nullptr;
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)wg), dim3(512), G::LDS_B, st, a);
  WSR_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// Forward of a z-tapless stride-1 conv with at most 16 outputs, planar fp32 output (see the file header).
// WSR_EUNSUPPORTED: shape without an instantiation - the caller runs the generic halo-tile kernel.
int wsr_conv_slide_fwd(const unsigned short* in, int in_ctot, int in_off, int C, const unsigned short* wfrag, float* out,
                       int N, int B, int X, int Y, int Z, int KX, int KY, int px, int py, const float* bias,
                       const void* zero16, hipStream_t st) {
  if (C % 16 || in_ctot % 8 || in_off % 8 || N < 1 || N > 16) return WSR_EUNSUPPORTED;
  if ((long)Y * Z * in_ctot >= (1l << 29)) return WSR_EUNSUPPORTED;  // 32-bit byte offsets inside an x-plane, below CS_OOB
  if ((long)16 * X * Y * Z >= (1l << 31)) return WSR_EUNSUPPORTED;     // 32-bit element offsets inside one item's output
  if ((long)B * Y * Z < 64 || X < 8) return WSR_EUNSUPPORTED;        // tiny volumes stay on the small-tile kernels
  CsArgs a{};
  a.in = in; a.wf = wfrag; a.out = out; a.zero16 = zero16; a.bias = bias;
  a.B = B; a.X = X; a.Y = Y; a.Z = Z; a.in_ctot = in_ctot; a.in_off = in_off; a.N = N; a.px = px; a.py = py;
  if (KX == 5 && KY == 5 && C == 144) return launch_slide_fwd<5, 5, 18>(a, st);
  if (KX == 3 && KY == 3 && C == 144) return launch_slide_fwd<3, 3, 18>(a, st);
  return WSR_EUNSUPPORTED;
}

// ================================================================================================================
//   input gradient (wsr_conv_slide_dgrad):  dx[v, c] = mask(h[v, c]) * scale[b, c] * sum_{kx,ky,n} dy[v + (kx,ky) - pad', n] * wT[c, (kx,ky), n]
//
// Short reduction (taps x 16 channels of dy), wide result (C = 144): the launch moves the C-channel HR tensor twice
// (mask source in, gradient out) and reads a 16-channel one.  Same sliding scheme: a workgroup owns a 16 x 4 column
// and walks x.  The dy planes are tiny (20 rows x 4 levels x 32 B) and ring through LDS eight deep; the filter
// (transposed, tap-flipped fragments of the tile kernels) lives in REGISTERS - a wave owns 2-3 of the nine 16-channel
// output tiles for two of the plane's four voxel tiles, all 13 K-steps (156 VGPRs).  The mask source (the saved
// output of hr_convs[0], reference Generator_3D_Resnet_ESRGAN.py:95-104 LeakyReLU + Dropout3d) arrives by LDS-DMA two
// planes ahead; results are staged in LDS as whole 288-byte voxel rows and leave as 16-byte stores of full rows
// (the halo-tile kernel stored 8 bytes per lane at a 288-byte stride).  DMA is issued by waves 0-3, stores by waves
// 4-7: no wave waits for a store to learn that its DMA has landed (see the forward kernel).
namespace {

struct CdArgs {
  const unsigned short* dy;    // NDHWC bf16, 16 channels read at [dy_off, dy_off + 16)
  const unsigned short* wf;    // transposed fragment filter [1 chunk][tap pair][n-tile][lane][8]
  unsigned short* dx;          // NDHWC bf16
  const unsigned short* mask;  // NDHWC bf16 (saved forward output), window [mask_off, mask_off + C)
  const float* chan_scale;     // [B][C] or NULL
  const void* zero16;
  int B, X, Y, Z;
  int dy_ctot, dy_off, dx_ctot, dx_off, mask_ctot, mask_off;
  int px, py;                  // gather pads (K - 1 - pad of the forward conv)
  float alpha, slope;
  int nty, ntz, nseg, XS, blk;
};

// AH: rounds of DMA kept in flight beyond the one being issued (0: what round 3 shipped - a plane's DMA has one round,
// 2.2 us, to land; 1: two rounds - the mask ring grows to five planes, the dy ring is used to its eighth slot).
// Measured (round 4, WSR_CS_AH=1, same device, tools/bench_conv.py hr1z): 370.5 / 373.8 us against 359.8 / 355.3 - the
// launch is not waiting for its reads; twice the reads in flight only get in the way of the result stores.  Default 0.
template <int KX, int KY, int NT, int AH>
struct CdGeom {
  static constexpr int TY = 16, TZ = 4;
  static constexpr int PY = TY + KY - 1;
  static constexpr int DY_NP = PY * TZ * 2;                 // 16-byte pieces of a dy plane (2 per voxel)
  static constexpr int DY_NU = (DY_NP + 63) / 64;
  static constexpr int DY_B = DY_NP * 16;
  static constexpr int RD = 8;                              // dy ring slots (power of two)
  static constexpr int OC = NT * 2;                         // octets of the wide side
  static constexpr int ROWB = OC * 16;                      // bytes of a result / mask voxel row
  static constexpr int M_NP = TY * TZ * OC;                 // pieces of a mask / result plane
  static constexpr int M_NU = (M_NP + 63) / 64;
  static constexpr int M_B = M_NP * 16;
  static constexpr int RM = 4 + AH;                         // mask ring slots: planes i - 1 .. i + 2 + AH
  static constexpr int NKS = (KX * KY + 1) / 2;             // K-steps: tap pairs x 16 channels
  static constexpr int TN = NT / 4;                         // n-tiles per wave held in registers (four N groups)
  static constexpr int NX = NT - 4 * TN;                    // left-over n-tile (0 or 1): N group 0, filter in LDS
  static constexpr int NU = M_NU + DY_NU;                   // DMA units per plane
  static constexpr int OFF_DY = 0, OFF_MASK = RD * DY_B, OFF_STAGE = OFF_MASK + RM * M_B;
  static constexpr int OFF_SC = OFF_STAGE + 2 * M_B;       // per-channel factors (NT * 16 floats)
  static constexpr int OFF_WX = OFF_SC + NT * 16 * 4;      // filter fragments of the left-over n-tile (NKS KB)
  static constexpr int LDS_B = OFF_WX + NX * NKS * 1024;
};

template <int KX, int KY, int NT, int AH>
__global__ __launch_bounds__(512) void conv_slide_dgrad_kernel(const CdArgs a) {
  using G = CdGeom<KX, KY, NT, AH>;
  constexpr int TY = G::TY, TZ = G::TZ, DY_NP = G::DY_NP, DY_NU = G::DY_NU, DY_B = G::DY_B, RD = G::RD, OC = G::OC;
  constexpr int ROWB = G::ROWB, M_NP = G::M_NP, M_NU = G::M_NU, M_B = G::M_B, RM = G::RM, NKS = G::NKS, TN = G::TN;
  constexpr int NU = G::NU, NX = G::NX;
  static_assert(NX <= 1, "one left-over n-tile at most");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ng = wave & 3, mh = wave >> 2;  // N group, m-tile half
  const int fr = lane & 15, fg = lane >> 4;
  // n-tiles of N group g: [g TN, (g + 1) TN) with the filter in registers.  The left-over tile NT - 1 (NT = 9: 144
  // channels) is shared out one voxel tile per wave to waves 0-3 - one per SIMD, so that every SIMD carries the same
  // number of MFMAs (with both of its voxel tiles on N group 0 that group's SIMD had 1.5 x the work and set the pace) -
  // and its filter fragments are read from LDS: three tiles of fragments per wave (156 VGPRs) left the compiler 4
  // registers short, and a spilled DMA base pointer is reloaded behind a vmcnt(0) every plane.
  const int nt_lo = ng * TN;
  const bool extra = NX > 0 && mh == 0;  // waves 0 .. 3: the left-over n-tile of voxel tile `ng`

  unsigned bid = (unsigned)xcd_remap(blockIdx.x, gridDim.x);
  int ty, tz;
  if (a.blk) {
    const unsigned within = bid & 31u;
    unsigned r = bid >> 5;
    const unsigned nzb = (unsigned)a.ntz >> 3, nyb = (unsigned)a.nty >> 2;
    const unsigned zb = r % nzb; r /= nzb;
    const unsigned yb = r % nyb; r /= nyb;
    ty = (int)(yb * 4 + (within >> 3));
    tz = (int)(zb * 8 + (within & 7u));
    bid = r;
  } else {
    tz = (int)(bid % (unsigned)a.ntz); bid /= (unsigned)a.ntz;
    ty = (int)(bid % (unsigned)a.nty); bid /= (unsigned)a.nty;
  }
  const int seg = (int)(bid % (unsigned)a.nseg);
  const int b = (int)(bid / (unsigned)a.nseg);
  const int y0 = ty * TY, z0 = tz * TZ;
  const int x_begin = seg * a.XS;
  const int nplanes = min(a.XS, a.X - x_begin);

  typedef __attribute__((address_space(3))) char* lptr_t;
  const unsigned lds0 = (unsigned)(unsigned long)(lptr_t)smem;
  char* const dyr = smem + G::OFF_DY;
  char* const mkr = smem + G::OFF_MASK;
  char* const stg = smem + G::OFF_STAGE;

  // ---- DMA units of waves 0-3: units [0, M_NU) = mask plane, [M_NU, NU) = dy plane; result units of waves 4-7 ------
  // (one offset table for both roles: a wave has one of them).  DMA goes through buffer descriptors (cs_bufdma16):
  // rows outside the tensor and planes past the segment read as zeros without any address arithmetic.
  const bool dma_wave = wave < 4, st_wave = !dma_wave;
  const int w4 = wave & 3;
  constexpr int UPW = (NU + 3) / 4, SPW = (M_NU + 3) / 4;
  unsigned uoff[UPW];   // byte offset inside an x-plane of the tensor the unit belongs to, or CS_OOB
  unsigned udst[UPW];   // LDS byte offset of the unit inside its plane (wave-uniform)
#pragma unroll
  for (int k = 0; k < UPW; ++k) {
    const int u = w4 + 4 * k;
    unsigned off = CS_OOB;
    unsigned dst = 0;
    if (u < M_NU) {  // a mask unit (DMA waves) or the result unit of the same voxels (store waves)
      const int p0 = u * 64 < M_NP - 64 ? u * 64 : M_NP - 64;
      // (a shifted last unit overlaps its predecessor: the overlap is fetched / stored twice with the same bytes)
      dst = (unsigned)p0 * 16u;
      const int p = p0 + lane;
      const int vox = p / OC, o = p - vox * OC;
      const int gy = y0 + vox / TZ, gz = z0 + vox % TZ;
      if (gy < a.Y && gz < a.Z)
        off = 2u * (unsigned)(dma_wave ? (gy * a.Z + gz) * a.mask_ctot + a.mask_off + o * 8
                                       : (gy * a.Z + gz) * a.dx_ctot + a.dx_off + o * 8);
    } else if (dma_wave && u < NU) {
      const int ud = u - M_NU;
      const int p0 = ud * 64 < DY_NP - 64 ? ud * 64 : DY_NP - 64;
      dst = (unsigned)p0 * 16u;
      const int p = p0 + lane;
      const int vox = p >> 1, o = p & 1;
      const int gy = y0 - a.py + vox / TZ, gz = z0 + vox % TZ;
      if ((unsigned)gy < (unsigned)a.Y && gz < a.Z) off = 2u * (unsigned)((gy * a.Z + gz) * a.dy_ctot + a.dy_off + o * 8);
    }
    uoff[k] = off;
    udst[k] = __builtin_amdgcn_readfirstlane(dst);
  }
  const long mask_plane = (long)a.Y * a.Z * a.mask_ctot, dy_plane = (long)a.Y * a.Z * a.dy_ctot;
  // descriptors of mask plane xm (output plane index relative to x_begin) and dy plane xd (input plane index
  // relative to x_begin - px); a plane that is not wanted or outside the tensor has zero records (reads as zeros)
  auto mask_srd = [&](int xm, bool wanted) __attribute__((always_inline)) {
    return cs_make_srd(a.mask + ((long)b * a.X + (wanted ? x_begin + xm : 0)) * mask_plane,
                       wanted ? (unsigned)(mask_plane * 2) : 0u);
  };
  auto dy_srd = [&](int xd, bool wanted) __attribute__((always_inline)) {
    const int xi = x_begin - a.px + xd;
    const bool xin = wanted && (unsigned)xi < (unsigned)a.X;
    return cs_make_srd(a.dy + ((long)b * a.X + (xin ? xi : 0)) * dy_plane, xin ? (unsigned)(dy_plane * 2) : 0u);
  };
  // DMA instruction k of this wave for (mask plane xm, dy plane xd)
  auto issue_unit = [&](int k, const cs_srd_t& sm, int xm, const cs_srd_t& sd, int xd) __attribute__((always_inline)) {
    const int u = w4 + 4 * k;
    if (u < M_NU)
      cs_bufdma16(sm, uoff[k], __builtin_amdgcn_readfirstlane(lds0 + G::OFF_MASK + ((unsigned)(xm + RM) % (unsigned)RM) * M_B + udst[k]));
    else if (u < NU)
      cs_bufdma16(sd, uoff[k], __builtin_amdgcn_readfirstlane(lds0 + G::OFF_DY + (unsigned)(xd & (RD - 1)) * DY_B + udst[k]));
  };

  const long dx_plane = (long)a.Y * a.Z * a.dx_ctot;

  // ---- filter fragments of this wave's n-tiles, all K-steps, in registers ------------------------------------
  uint4 wreg[NKS][TN];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      wreg[ks][j] = *reinterpret_cast<const uint4*>(a.wf + ((size_t)(ks * NT + nt_lo + j) * 64 + lane) * 8);
  char* const wxl = smem + G::OFF_WX;
  if constexpr (NX > 0) {
    for (int idx = t; idx < NKS * 64; idx += 512)
      *reinterpret_cast<uint4*>(wxl + idx * 16) =
          *reinterpret_cast<const uint4*>(a.wf + ((size_t)((idx >> 6) * NT + NT - 1) * 64 + (idx & 63)) * 8);
  }
  // per-channel factors (Dropout3d keep factor x alpha) wait in LDS: 4 per lane and n-tile, read when needed
  float* const scl = reinterpret_cast<float*>(smem + G::OFF_SC);
  for (int c = t; c < NT * 16; c += 512) scl[c] = (a.chan_scale ? a.chan_scale[(long)b * (NT * 16) + c] : 1.f) * a.alpha;
  // dy fragment of K-step ks: lane group fg holds tap 2 ks + (fg >> 1), channel octet fg & 1, voxel fr of the m-tile.
  // Both taps of a K-step are compile-time constants, so the fragment address is the lane's base + one of two
  // scalars (ring slot of the tap's column, row offset of the tap) picked by the lane group - no tables in registers.
  const int lane_base = (2 * mh * 16 + fr) * 32 + (fg & 1) * 16;  // (m-tile 2 mh; the second one is + 16 voxels)
  const int lane_extra = (ng - 2 * mh) * 16 * 32;                  // (voxel tile ng of the left-over n-tile, from lane_base)
  const bool tap_hi = (fg >> 1) != 0;

  // ---- prologue: dy planes 0 .. KX + AH, mask planes 0 .. 1 + AH ---------------------------------------------------
  const int last_dy = nplanes + KX - 2;
  if (dma_wave) {
    const cs_srd_t none = cs_make_srd(a.mask, 0u);
#pragma unroll
    for (int d = 0; d <= KX + AH; ++d) {
      const cs_srd_t sd = dy_srd(d, d <= last_dy);
#pragma unroll
      for (int k = 0; k < UPW; ++k)
        if (w4 + 4 * k >= M_NU) issue_unit(k, none, 0, sd, d);
    }
#pragma unroll
    for (int m = 0; m < 2 + AH; ++m) {
      const cs_srd_t sm = mask_srd(m, m < nplanes);
#pragma unroll
      for (int k = 0; k < UPW; ++k)
        if (w4 + 4 * k < M_NU) issue_unit(k, sm, m, none, 0);
    }
  }
  cs_dma_wait();
  __syncthreads();

  // ---- main loop ---------------------------------------------------------------------------------------------------
  // Round i contracts plane i and, BETWEEN its K-steps, (a) issues the DMA of mask plane i + 2 / dy plane i + KX + 1
  // (waves 0-3; the mask ring holds planes i - 1 .. i + 2), (b) masks, scales, rounds and stages plane i - 1 from the accumulators kept from the previous round
  // (vector work under the MFMAs; done in a block behind them it cost a quarter of the launch), (c) stores the staged
  // rows of plane i - 2 (waves 4-7).  The body is specialised per wave role and straight-line: the same loop with
  // ~80 scalar branches per plane (role / first / last tests inside the unrolled K loop) spent 0.85 us per plane on
  // them alone.  Planes past the end run with zero-record descriptors and rows that are never stored.
  const int nw = dma_wave ? (NU - w4 + 3) / 4 : 0;  // DMA instructions this wave issues per plane
  constexpr int NWMAX = (NU + 3) / 4;
  static_assert(UPW <= NKS && 2 * SPW <= NKS && (TN + NX) * 2 <= NKS, "one side job per K-step");
  f32x4_t accp[2][TN + NX];  // accumulators of the plane contracted in the previous round
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int j = 0; j < TN + NX; ++j) accp[m][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto round = [&](int i, auto dmaw_c) __attribute__((always_inline)) {
    constexpr bool DMAW = decltype(dmaw_c)::value;  // waves 0-3: DMA + the left-over n-tile; waves 4-7: stores
    const cs_srd_t sm = mask_srd(i + 2 + AH, i + 2 + AH < nplanes), sd = dy_srd(i + KX + 1 + AH, i + KX + 1 + AH <= last_dy);
    // rows staged in round i - 1 (plane i - 2) leave now
    const int ip = i - 2;
    char* const st_base = reinterpret_cast<char*>(a.dx + ((long)b * a.X + x_begin + (ip > 0 ? ip : 0)) * dx_plane);
    const char* const sg_out = stg + (ip & 1) * M_B;
    const bool st_ok = ip >= 0 && ip < nplanes;
    uint4 sreg = make_uint4(0u, 0u, 0u, 0u);
    // plane i - 1 is masked / staged now
    const char* const mk = mkr + ((unsigned)(i - 1 + RM) % (unsigned)RM) * M_B;
    char* const sg_in = stg + ((i - 1) & 1) * M_B;
    f32x4_t acc[2][TN + NX];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int j = 0; j < TN + NX; ++j) acc[m][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int D = 2;
    uint4 xa[D], xb[D], wx[D], xc[D];
#define CD_FRAG_ADDR(ks)                                                                                        \
  (dyr + lane_base +                                                                                            \
   (tap_hi ? ((i + ((2 * (ks) + 1 < KX * KY - 1 ? 2 * (ks) + 1 : KX * KY - 1) / KY)) & (RD - 1)) * DY_B +        \
                 ((2 * (ks) + 1 < KX * KY - 1 ? 2 * (ks) + 1 : KX * KY - 1) % KY) * TZ * 32                       \
           : ((i + ((2 * (ks) < KX * KY - 1 ? 2 * (ks) : KX * KY - 1) / KY)) & (RD - 1)) * DY_B +                \
                 ((2 * (ks) < KX * KY - 1 ? 2 * (ks) : KX * KY - 1) % KY) * TZ * 32))
#define CD_FETCH(ks)                                                                                            \
  do {                                                                                                          \
    const char* p_ = CD_FRAG_ADDR(ks);                                                                          \
    xa[(ks) % D] = *reinterpret_cast<const uint4*>(p_);                                                         \
    xb[(ks) % D] = *reinterpret_cast<const uint4*>(p_ + 16 * 32);                                               \
    if constexpr (NX > 0 && DMAW) {                                                                             \
      wx[(ks) % D] = *reinterpret_cast<const uint4*>(wxl + (ks) * 1024 + lane * 16);                             \
      xc[(ks) % D] = *reinterpret_cast<const uint4*>(p_ + lane_extra);                                           \
    }                                                                                                           \
  } while (0)
#pragma unroll
    for (int ks = 0; ks < D && ks < NKS; ++ks) CD_FETCH(ks);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        mma_chunk<BF16>(acc[0][j], wreg[ks][j], xa[ks % D]);
        mma_chunk<BF16>(acc[1][j], wreg[ks][j], xb[ks % D]);
      }
      if constexpr (NX > 0 && DMAW) mma_chunk<BF16>(acc[0][TN], wx[ks % D], xc[ks % D]);
      if (ks + D < NKS) CD_FETCH(ks + D);
      __builtin_amdgcn_sched_barrier(0);
      // (a) one DMA instruction
      if constexpr (DMAW) {
        if (ks < UPW) issue_unit(ks, sm, i + 2 + AH, sd, i + KX + 1 + AH);
      }
      // (b) one (voxel tile, n-tile) of plane i - 1: mask, scale, round, stage as [voxel][ROWB] rows
      if (ks < 2 * (TN + NX)) {
        const int m = ks / (TN + NX), j = ks % (TN + NX);
        if (j < TN || (NX > 0 && DMAW && m == 0)) {
          const int nt = j < TN ? nt_lo + j : NT - 1;
          const int mt = j < TN ? 2 * mh + m : ng;
          const int ro = (mt * 16 + fr) * ROWB + (nt * 16 + 4 * fg) * 2;
          const uint2 y = *reinterpret_cast<const uint2*>(mk + ro);
          const float4 s4 = *reinterpret_cast<const float4*>(scl + nt * 16 + 4 * fg);
          float4 o4 = make_float4(accp[m][j][0] * s4.x, accp[m][j][1] * s4.y, accp[m][j][2] * s4.z, accp[m][j][3] * s4.w);
          o4.x *= (short)(y.x & 0xFFFFu) > 0 ? 1.f : a.slope;  // bf16 sign test on the raw bits: y > 0
          o4.y *= (int)y.x > 0xFFFF ? 1.f : a.slope;
          o4.z *= (short)(y.y & 0xFFFFu) > 0 ? 1.f : a.slope;
          o4.w *= (int)y.y > 0xFFFF ? 1.f : a.slope;
          uint2 u;
          u.x = (unsigned)f2bf(o4.x) | ((unsigned)f2bf(o4.y) << 16);
          u.y = (unsigned)f2bf(o4.z) | ((unsigned)f2bf(o4.w) << 16);
          *reinterpret_cast<uint2*>(sg_in + ro) = u;
        }
      }
      // (c) result unit ks / 2 of plane i - 2: LDS read on even steps, store on odd ones
      if constexpr (!DMAW) {
        if (ks < 2 * SPW) {
          const int k = ks >> 1;
          if (w4 + 4 * k < M_NU) {
            if (!(ks & 1)) sreg = *reinterpret_cast<const uint4*>(sg_out + udst[k] + lane * 16);
            else if (st_ok && uoff[k] != CS_OOB) *reinterpret_cast<uint4*>(st_base + uoff[k]) = sreg;
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#undef CD_FETCH
#undef CD_FRAG_ADDR
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int j = 0; j < TN + NX; ++j) accp[m][j] = acc[m][j];
    // the DMA issued 1 + AH rounds ago has landed; the younger rounds' stay in flight (every round issues the same
    // number of instructions per wave: planes past the end go out with zero-record descriptors)
    if constexpr (DMAW) {
      if (nw == NWMAX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((1 + AH) * NWMAX) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((1 + AH) * (NWMAX - 1)) : "memory");
    }
    __syncthreads();
  };
  // rounds 0 .. nplanes + 1: the last two only drain the pipeline (stage the last plane, store the last two)
  if (dma_wave) {
    for (int i = 0; i < nplanes + 2; ++i) round(i, std::true_type{});
  } else {
    for (int i = 0; i < nplanes + 2; ++i) round(i, std::false_type{});
  }
  cs_dma_wait();  // (zero-record DMA issued for planes past the end must not outlive the workgroup)
}

template <int KX, int KY, int NT, int AH>
int launch_slide_dgrad(CdArgs& a, hipStream_t st) {
  using G = CdGeom<KX, KY, NT, AH>;
  static_assert(G::LDS_B <= 160 * 1024, "rings + staging exceed the LDS");
  static_assert(G::DY_NP >= 64 && G::M_NP >= 64, "plane smaller than one DMA unit");
  static_assert((G::NU + 3) / 4 <= 6, "the counted DMA wait covers at most six instructions per plane and wave");
  static_assert(KX + 2 + AH <= G::RD, "dy ring too short");
  static_assert((1 + AH) * ((G::NU + 3) / 4) < 64, "vmcnt is a 6-bit counter");
  auto kern = conv_slide_dgrad_kernel<KX, KY, NT, AH>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       G::LDS_B);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  a.nty = (a.Y + G::TY - 1) / G::TY;
  a.ntz = (a.Z + G::TZ - 1) / G::TZ;
  const long cols = (long)a.B * a.nty * a.ntz;
  int nseg = (int)((256 + cols - 1) / cols);
  if (nseg > a.X / 8) nseg = a.X / 8;
  if (nseg < 1) nseg = 1;
  a.XS = (a.X + nseg - 1) / nseg;
  a.nseg = (a.X + a.XS - 1) / a.XS;
  a.blk = (a.nty % 4 == 0 && a.ntz % 8 == 0) ? 1 : 0;
  const long wg = cols * a.nseg;
  if (wg >= (1l << 31)) return WSR_EUNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3((unsigned)wg), dim3(512), G::LDS_B, st, a);
  WSR_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// Input gradient of a z-tapless stride-1 conv whose OUTPUT side has at most 16 channels, with the LeakyReLU (+
// Dropout3d keep factor) backward of the layer below in the epilogue; C = channels of dx.  `px`, `py` are the pads of
// the gather over dy (K - 1 - pad).  WSR_EUNSUPPORTED: no instantiation - the caller runs the halo-tile kernel.
int wsr_conv_slide_dgrad(const unsigned short* dy, int dy_ctot, int dy_off, int red, const unsigned short* wfrag_t,
                         unsigned short* dx, int dx_ctot, int dx_off, int C, int B, int X, int Y, int Z, int KX, int KY,
                         int px, int py, float alpha, const wsr_lrelu_mask_t* mask, const void* zero16, hipStream_t st) {
  if (!mask || !mask->y || mask->c0 != 0 || mask->c1 != C || red != 16) return WSR_EUNSUPPORTED;
  if (dy_ctot % 8 || dy_off % 8 || dx_ctot % 8 || dx_off % 8 || mask->y_ctot % 8 || mask->y_off % 8) return WSR_EUNSUPPORTED;
  if ((long)Y * Z * dx_ctot >= (1l << 29) || (long)Y * Z * mask->y_ctot >= (1l << 29) || (long)Y * Z * dy_ctot >= (1l << 29))
    return WSR_EUNSUPPORTED;  // 32-bit byte offsets inside an x-plane, below CS_OOB
  if ((long)B * Y * Z < 64 || X < 8) return WSR_EUNSUPPORTED;
  CdArgs a{};
  a.dy = dy; a.wf = wfrag_t; a.dx = dx; a.mask = (const unsigned short*)mask->y; a.chan_scale = mask->chan_scale;
  a.zero16 = zero16;
  a.B = B; a.X = X; a.Y = Y; a.Z = Z;
  a.dy_ctot = dy_ctot; a.dy_off = dy_off; a.dx_ctot = dx_ctot; a.dx_off = dx_off;
  a.mask_ctot = mask->y_ctot; a.mask_off = mask->y_off;
  a.px = px; a.py = py; a.alpha = alpha; a.slope = mask->slope;
  if (KX == 5 && KY == 5 && C == 144)
    return WSR_ENV_INT("WSR_CS_AH", 0) ? launch_slide_dgrad<5, 5, 9, 1>(a, st) : launch_slide_dgrad<5, 5, 9, 0>(a, st);
  // (other filter extents - no shipped configuration has them - stay on the halo-tile kernel)
  return WSR_EUNSUPPORTED;
}
