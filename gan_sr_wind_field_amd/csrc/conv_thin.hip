// Memory-bound 3x3x3 convolutions with a thin INPUT side (<= 16 reduction channels), bf16: a workgroup SLIDES along x.
//
// SURVEY 8(d)'s memory-bound conv3d set besides the last conv (conv_slide.hip): the terrain branch of the generator
// (reference Generator_3D_Resnet_ESRGAN.py:111-119: 3x3x3 1 -> 16 + LeakyReLU, 3x3x3 16 -> 16 into the concat, at HR
// resolution), its feature conv (:78-85, 4 -> 128) and the discriminator's first conv (Discriminator_3D.py:67-75,
// 3 -> 32 + LeakyReLU).  Their arithmetic is a few GFLOP per launch; the launch is the time to read the input once
// and write the output once.  On the halo-tile kernel (conv_tile_impl.h) every 512-voxel tile paid a DMA round trip,
// a dozen K-steps and an epilogue one after the other: 0.13-0.22 of the HBM rate (profiles/r03_h_hbm_kernels.txt).
//
//   y[v, n] = act(bias[n] + sum_{tap, c} x[v + tap - 1, c] * w[n, tap, c]) * alpha        K = 27 taps x CP channels
//
// A workgroup (8 waves) owns a column of the volume - TY rows x TZ levels, every x of its segment - and streams the
// x-planes of the input (rows and levels WITH their halo, all CP channels) through a ring of NB LDS buffers by
// buffer-descriptor LDS-DMA, D = NB - 1 planes in flight: halo voxels and planes outside the tensor read as zeros
// with no address arithmetic.  Input-stationary: a fragment of plane p (one (ky, kz) tap group of one 16-voxel
// z-run) is read from LDS ONCE and multiplied into the accumulators of the three output planes p + 1 - kx; the
// accumulator set of output plane p - 1 is complete after plane p, the one of plane p + 1 starts from zero with its
// first MFMA.  FOUR sets rotate: the completed one leaves through the epilogue (bias, LeakyReLU, stores into the NDHWC
// channel window) one iteration LATER, interleaved with the next plane's MFMAs, so that the matrix pipe and the store
// path work at the same time (with three sets and the epilogue behind the MFMAs the two times simply added up).  The
// filter never touches LDS: every wave holds the fragments of its n-tiles in registers for the whole launch (taps of
// one K-step share their kx: 3 x 5 K-steps of 2 taps x 16 channels, or 3 x 3 K-steps of 4 taps x 8 channels), read
// from the tile kernels' fragment-order filter (wsr_pack_filter_frag) - no second packed format.
//
// Same entry points as the other tile kernels: wsr_conv3d_fwd_tile / wsr_conv3d_dgrad_tile dispatch here first
// (conv_tile.hip) and fall through on WSR_EUNSUPPORTED.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) int ct3_srd_t;
typedef __attribute__((ext_vector_type(2))) unsigned ct3_u2;
typedef __attribute__((ext_vector_type(4))) unsigned ct3_u4;

template <int I, int N, class F>
__device__ __forceinline__ void ct3_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ct3_static_for<I + 1, N>(f);
  }
}

// LDS-DMA through a buffer descriptor: LDS[lds_addr + 16*lane] <- 16 bytes at srd.base + voff, ZEROS where voff is
// outside [0, srd.num_records).  (s_nop 4: SALU-write -> VMEM-read hazard of M0 / the descriptor inside the statement.)
__device__ __forceinline__ void ct3_bufdma16(ct3_srd_t srd, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %2, %1, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(srd), "v"(voff), "s"(lds_addr)
      : "memory");
}
constexpr unsigned CT3_OOB = 0x7FFFFFF0u;  // a byte offset no plane reaches (the launch checks): reads as zeros

// wait until at most N of this wave's vector-memory operations are still in flight (N is an immediate: the loop bodies
// are straight-line - a run-time count through a jump table, the ring index by a division and the unit guards cost
// ~100 scalar branches per plane, 1.1 us of the 1.3 us an iteration took with everything else switched off)
template <int N> __device__ __forceinline__ void ct3_wait() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct Ct3Args {
  const unsigned short* in;  // NDHWC bf16, window [in_off, in_off + CP) of in_ctot channels
  const unsigned short* wf;  // tile-kernel fragment filter of the conv (TPK = 32 / CP ... see the kernel)
  unsigned short* out;       // NDHWC bf16, window [out_off, out_off + N) of out_ctot channels
  const float* bias;         // [N] or NULL
  float alpha, slope;
  int act;
  int B, X, Y, Z;
  int in_ctot, in_off, out_ctot, out_off, N, NT_total;
  int nty, ntz, nseg, XS;    // tiles along y and z, segments along x and their length
  int ablate;                // -DWSR_CT3_ABL_RT builds, WSR_CT3_ABL (timing only, wrong results): 1 no DMA after the first
                             // planes, 2 no MFMAs, 4 no stores, 8 no LDS fragment reads, 16 no barrier
};

#ifdef WSR_CT3_ABL_RT
#define CT3_ABL(bit) (a.ablate & (bit))
#else
#define CT3_ABL(bit) false
#endif

// Compile-time geometry of an instantiation.  CP: reduction channels as stored (8 or 16).  WN waves share a plane's
// m-tiles and split the n-tiles (NTW each); 8 / WN wave rows split the m-tiles (MT each); the column is TY rows x TZ
// levels (TZ = 16 or 32), its planes carry a one-voxel halo in y and z.
// PS ("paired stores", NTW = 1 and MT even): a lane holds 4 channels of one voxel per m-tile - an 8-byte store, and 8-byte
// stores run at 0.54-0.70 of the 16-byte rate (MI355X_MICROARCH.md).  With the n-tile's rows dealt to the lane groups as
// channel blocks (0, 2, 1, 3) one v_permlane32_swap per dword between the results of an m-tile PAIR leaves lane groups 0 / 1
// with channels 0-7 / 8-15 of the first m-tile's voxel and groups 2 / 3 with the same of the second's: ONE 16-byte store per
// lane and pair (the guide's T21).
template <int CP, int WN, int NTW, int MT, int TZ, int WAVES_, bool PS = false>
struct Ct3Geom {
  static constexpr int WAVES = WAVES_, WM = WAVES / WN;
  static constexpr int MTILES = WM * MT;
  static constexpr int TY = MTILES * 16 / TZ;
  static constexpr int LY = TY + 2, LZ = TZ + 2;
  static constexpr int TPK = 32 / CP;             // taps per K-step (2: 16 channels each; 4: 8 channels each)
  static constexpr int PPV = CP / 8;              // 16-byte pieces per voxel
  static constexpr int RB = CP * 2;               // bytes per voxel row in LDS
  static constexpr int SK = (9 + TPK - 1) / TPK;  // K-steps per kx: the nine (ky, kz) taps in groups of TPK
  static constexpr int NP = LY * LZ * PPV;        // 16-byte pieces of a plane
  static constexpr int NU = (NP + 63) / 64;       // 1 KB DMA units of a plane
  static constexpr int DUW = (NU + WAVES - 1) / WAVES;  // units of a plane per wave: DUW for waves < NXW, DUW - 1 beyond
  static constexpr int NXW = NU - WAVES * (DUW - 1);
  static constexpr int STRIDE = NU * 1024;        // bytes between plane buffers
  static constexpr int NB_FIT = (WAVES == 8 ? 64 * 1024 : 39 * 1024) / STRIDE;  // LDS for 16 waves per CU
  static constexpr int NB = NB_FIT > 8 ? 8 : NB_FIT;   // plane buffers
  static constexpr int D = NB - 1;                // planes in flight
  static constexpr int SPP = PS ? MT / 2 : MT;    // store instructions per wave and output plane (8 or 16 B per lane)
  static_assert(!PS || (NTW == 1 && MT % 2 == 0), "paired stores: one n-tile, an even number of m-tiles");
  static_assert(TY >= 1 && TY * TZ == MTILES * 16, "the plane's m-tiles must fill TY x TZ");
  static_assert(NB >= 3, "at least two planes in flight");
  static_assert((D - 1) * DUW + D * SPP <= 63, "counted waits: vmcnt is a 6-bit counter");
};

template <int CP, int WN, int NTW, int MT, int TZ, int WAVES, bool PS = false>
__global__ __launch_bounds__(WAVES * 64) void conv_thin3_kernel(const Ct3Args a) {
  using G = Ct3Geom<CP, WN, NTW, MT, TZ, WAVES, PS>;
  constexpr int WM = G::WM, TY = G::TY, LZ = G::LZ, TPK = G::TPK, PPV = G::PPV, RB = G::RB, SK = G::SK;
  constexpr int NP = G::NP, DUW = G::DUW, NXW = G::NXW, NB = G::NB, D = G::D, SPP = G::SPP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = wave % WN, wm = wave / WN;
  const int fr = lane & 15, fg = lane >> 4;

  // ---- which column ------------------------------------------------------------------------------------
  unsigned bid = (unsigned)xcd_remap(blockIdx.x, gridDim.x);
  const int tz = (int)(bid % (unsigned)a.ntz); bid /= (unsigned)a.ntz;
  const int ty = (int)(bid % (unsigned)a.nty); bid /= (unsigned)a.nty;
  const int seg = (int)(bid % (unsigned)a.nseg);
  const int b = (int)(bid / (unsigned)a.nseg);
  const int y0 = ty * TY, z0 = tz * TZ;
  const int x_begin = seg * a.XS;
  const int nout = min(a.XS, a.X - x_begin);  // output planes of this workgroup (>= 1 by construction)
  const int nin = nout + 2;                   // input planes: x_begin - 1 .. x_begin + nout

  // ---- filter fragments -> registers ---------------------------------------------------------------------
  // K-step j of tap column kx holds taps (kx, cb), cb = TPK*j + sub, sub = fg / PPV (the lane's tap within the
  // K-step), octet fg % PPV; taps past the ninth carry zero weights.  Packed layout (conv_tile.hip):
  // [K-step = tap / TPK][n-tile][lane = ((tap % TPK) * PPV + octet) * 16 + row][8 channels]
  const int sub = fg / PPV, oct = fg % PPV;
  uint4 W[3][SK][NTW];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int j = 0; j < SK; ++j)
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        // row fr of this wave's n-tile n = output channel ch.  NTW = 2: the rows of the tile pair are interleaved
        // (tile n <- channels 8g + 4n + 0..3 of the pair's 32, g = row / 4), so that a lane ends up with 8 CONSECUTIVE
        // channels of its voxel: one 16-byte store, whole 64-byte runs per voxel (two 8-byte stores per voxel wrote
        // every line half by half: the stores were 2/3 of the discriminator's first conv)
        const int cb = TPK * j + sub;
        // (PS: row block g of the n-tile <- channel block (0, 2, 1, 3)[g], see Ct3Geom)
        const int ch = NTW == 2 ? wn * 32 + 8 * (fr >> 2) + 4 * n + (fr & 3)
                                : (PS ? wn * 16 + 4 * ((((fr >> 2) & 1) << 1) | (fr >> 3)) + (fr & 3) : (wn * NTW + n) * 16 + fr);
        uint4 w = make_uint4(0u, 0u, 0u, 0u);
        if (cb < 9 && ch < a.NT_total * 16) {
          const int tap = kx * 9 + cb;
          w = *reinterpret_cast<const uint4*>(a.wf + ((size_t)(tap / TPK) * a.NT_total + (ch >> 4)) * 512 +
                                              (((tap % TPK) * PPV + oct) * 16 + (ch & 15)) * 8);
        }
        W[kx][j][n] = w;
      }

  // ---- per-lane LDS geometry -------------------------------------------------------------------------------
  // m-tile mt = wm + WM*i of the plane: row ry = mt / (TZ/16), z-run zm = (mt % (TZ/16)) * 16; lane fr = voxel zm + fr
  constexpr int ZRUNS = TZ / 16;
  int hb[MT], mry[MT], mzm[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int mt = wm + WM * i;
    mry[i] = mt / ZRUNS;
    mzm[i] = (mt % ZRUNS) * 16;
    hb[i] = (mry[i] * LZ + mzm[i] + fr) * RB + oct * 16;
  }
  int toff[SK];
#pragma unroll
  for (int j = 0; j < SK; ++j) {
    int cb = TPK * j + sub;
    if (cb > 8) cb = 8;  // (zero weights: any landed voxel will do)
    toff[j] = ((cb / 3) * LZ + (cb % 3)) * RB;
  }

  // ---- DMA geometry: the plane-relative byte offset of every piece this lane fetches (the same for all planes) -----
  unsigned voff[DUW];
#pragma unroll
  for (int k = 0; k < DUW; ++k) {
    const int q = (wave + WAVES * k) * 64 + lane;
    unsigned off = CT3_OOB;
    if (q < NP) {
      const int h = q / PPV, part = q - h * PPV;
      const int r = h / LZ, c = h - r * LZ;
      const int gy = y0 - 1 + r, gz = z0 - 1 + c;
      if ((unsigned)gy < (unsigned)a.Y && (unsigned)gz < (unsigned)a.Z)
        off = (unsigned)(((gy * a.Z + gz) * a.in_ctot + a.in_off + 8 * part) * 2);
    }
    voff[k] = off;
  }
  typedef __attribute__((address_space(3))) char* lptr_t;
  const unsigned lds0 = (unsigned)(unsigned long)(lptr_t)smem;
  const unsigned plane_bytes = (unsigned)a.Y * a.Z * a.in_ctot * 2u;
  const unsigned long long in_base = (unsigned long long)a.in + (unsigned long long)b * a.X * plane_bytes;
  const bool extra_unit = wave < NXW;  // (wave-uniform) this wave issues DUW units per plane, the others DUW - 1
  // input plane index pi (0 .. nout + 1) = tensor plane x_begin - 1 + pi -> ring buffer `buf`
  auto issue_plane = [&](int pi, int buf) __attribute__((always_inline)) {
    const int p = x_begin - 1 + pi;
    const bool ok = pi < nin && (unsigned)p < (unsigned)a.X;
    const unsigned long long base = in_base + (unsigned long long)(ok ? p : 0) * plane_bytes;
    ct3_srd_t srd;
    srd.x = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
    srd.y = __builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xFFFFu));  // stride 0
    srd.z = __builtin_amdgcn_readfirstlane(ok ? (int)plane_bytes : 0);      // zero records: the plane reads as zeros
    srd.w = 0x00020000;
    const unsigned dst = lds0 + (unsigned)buf * (unsigned)G::STRIDE + (unsigned)wave * 1024u;
#pragma unroll
    for (int k = 0; k < DUW - 1; ++k) ct3_bufdma16(srd, voff[k], __builtin_amdgcn_readfirstlane(dst + k * WAVES * 1024));
    if (NXW == WAVES || extra_unit)
      ct3_bufdma16(srd, voff[DUW - 1], __builtin_amdgcn_readfirstlane(dst + (DUW - 1) * WAVES * 1024));
  };

#pragma unroll
  for (int pi = 0; pi < D; ++pi) issue_plane(pi, pi);

  // ---- epilogue operands --------------------------------------------------------------------------------------
  // lane (voxel fr, group fg) holds, per n-tile n, 4 channels starting at cbase(n): NTW = 1: 16*tile + 4*fg;
  // NTW = 2 (interleaved pair): 32*wn + 8*fg + 4*n - the two tiles together are channels 32*wn + 8*fg .. + 7
  float bb[NTW][4];
#pragma unroll
  for (int n = 0; n < NTW; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = (NTW == 2 ? wn * 32 + 8 * fg + 4 * n
                               : (PS ? wn * 16 + 4 * (((fg & 1) << 1) | (fg >> 1)) : (wn * NTW + n) * 16 + fg * 4)) + r;
      bb[n][r] = (a.bias && co < a.N) ? a.bias[co] : 0.f;
    }
  const float neg = a.act ? a.slope : 1.f;  // LeakyReLU as a select (no branch in the store path)
  // Output row of m-tile i in a plane: byte offset of the lane's first channel of voxel (y0 + ry, z0 + zm + fr) from the
  // plane's first byte.  Stores go through a buffer descriptor of the output PLANE: rows outside the tensor (and
  // channels past the last) carry an out-of-range offset and are dropped by the hardware - every wave issues the same
  // number of store instructions per plane, with no divergence, which is what lets the DMA waits be COUNTED (vmcnt is
  // one in-order counter for loads and stores).
  // (PS: after the swap lane groups 0 / 1 hold channels 0-7 / 8-15 of the pair's FIRST m-tile, groups 2 / 3 of its second)
  const int co0 = NTW == 2 ? wn * 32 + 8 * fg : (PS ? wn * 16 + 8 * (fg & 1) : wn * 16 + fg * 4);
  unsigned orow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int im = PS ? (i & ~1) + (fg >> 1) : i;  // (PS: only the even entries are used, one per pair)
    const int gy = y0 + mry[im], gz = z0 + mzm[im] + fr;
    orow[i] = (gy < a.Y && gz < a.Z && co0 < a.N) ? (unsigned)((((gy * a.Z + gz) * a.out_ctot) + a.out_off + co0) * 2)
                                                  : CT3_OOB;
  }
  const unsigned out_plane_bytes = (unsigned)a.Y * a.Z * a.out_ctot * 2u;
  const unsigned long long out_base =
      (unsigned long long)a.out + ((unsigned long long)b * a.X + (unsigned long long)(x_begin - 1)) * out_plane_bytes;

  f32x4_t acc[4][MT][NTW];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int n = 0; n < NTW; ++n) acc[s][i][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  int rb = 0;  // ring buffer of the plane being contracted; the DMA of plane pi + D goes to the one before it
  // epilogue + store of m-tile i of accumulator set S as output index oi (output plane x_begin - 1 + oi)
  unsigned held[2] = {0u, 0u};  // (PS) packed results of a pair's first m-tile
  auto store_tile = [&](auto sc, int i, const __amdgpu_buffer_rsrc_t& orsrc) __attribute__((always_inline)) {
    constexpr int S = decltype(sc)::value;
    unsigned o[NTW][2];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = acc[S][i][n][r] + bb[n][r];
        x = x > 0.f ? x : x * neg;
        v[r] = x * a.alpha;
      }
      o[n][0] = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
      o[n][1] = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    }
    if constexpr (PS) {
      if ((i & 1) == 0) {  // the pair's first m-tile waits for the second (the loops are unrolled: i is a constant)
        held[0] = o[0][0];
        held[1] = o[0][1];
      } else {
        auto r0 = __builtin_amdgcn_permlane32_swap(held[0], o[0][0], false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(held[1], o[0][1], false, false);
        ct3_u4 o4;
        o4.x = r0[0]; o4.y = r1[0]; o4.z = r0[1]; o4.w = r1[1];
        __builtin_amdgcn_raw_buffer_store_b128(o4, orsrc, (int)orow[i & ~1], 0, 0);
      }
    } else if constexpr (NTW == 2) {
      ct3_u4 o4;
      o4.x = o[0][0]; o4.y = o[0][1]; o4.z = o[1][0]; o4.w = o[1][1];
      __builtin_amdgcn_raw_buffer_store_b128(o4, orsrc, (int)orow[i], 0, 0);
    } else {
      ct3_u2 o2;
      o2.x = o[0][0]; o2.y = o[0][1];
      __builtin_amdgcn_raw_buffer_store_b64(o2, orsrc, (int)orow[i], 0, 0);
    }
  };
  auto out_rsrc = [&](int oi) __attribute__((always_inline)) {
    const unsigned long long base = out_base + (unsigned long long)oi * out_plane_bytes;
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), 0, (int)out_plane_bytes, 0x00020000);
  };
  // One input plane.  R = pi % 4 (compile time): plane pi feeds output index oi = pi + 1 - kx (output plane
  // x_begin - 1 + oi), kept in accumulator set oi % 4: kx = 0 -> set (R + 1) % 4, started from zero; kx = 1 -> set R;
  // kx = 2 -> set (R + 3) % 4, complete afterwards (oi = pi - 1).  Set (R + 2) % 4 was completed by the PREVIOUS plane
  // (oi = pi - 2): its epilogue and stores are interleaved with this plane's MFMAs, m-tile by m-tile (from pi = 3 on).
  auto plane = [&](int pi, auto rc) __attribute__((always_inline)) {
    constexpr int R = decltype(rc)::value;
    constexpr int S0 = (R + 1) % 4, S1 = R, S2 = (R + 3) % 4, SD = (R + 2) % 4;
    // The plane's DMA has landed when, of this wave's vector-memory operations, only the younger ones are still in
    // flight: the DMA units of the D - 1 planes behind it and the stores issued since - D * SPP of them once every one
    // of the D iterations before this one has stored (pi >= D + 3; before that the count ignores the stores: stricter
    // than necessary for a handful of iterations).  Then everybody's.
    if (pi >= D + 3) {
      if (NXW == WAVES || extra_unit) ct3_wait<(D - 1) * DUW + D * SPP>();
      else ct3_wait<(D - 1) * (DUW - 1) + D * SPP>();
    } else {
      if (NXW == WAVES || extra_unit) ct3_wait<(D - 1) * DUW>();
      else ct3_wait<(D - 1) * (DUW - 1)>();
    }
    if (!CT3_ABL(16)) __syncthreads();
    const int prev = rb == 0 ? NB - 1 : rb - 1;
    if (!CT3_ABL(1)) issue_plane(pi + D, prev);  // the buffer read in the previous iteration: free since the barrier
    const char* xs = smem + rb * G::STRIDE;
    rb = rb + 1 == NB ? 0 : rb + 1;
    const __amdgpu_buffer_rsrc_t orsrc = out_rsrc(pi - 2);
    const bool store = pi >= 3 && !CT3_ABL(4);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      uint4 xf[1][SK];  // (fragments of one m-tile at a time: two WGs per CU need <= 128 registers)
#pragma unroll
      for (int j = 0; j < SK; ++j)
        xf[0][j] = CT3_ABL(8) ? make_uint4(lane, 1u, 2u, 3u) : *reinterpret_cast<const uint4*>(xs + hb[i] + toff[j]);
      if (!CT3_ABL(2)) {
#pragma unroll
        for (int j = 0; j < SK; ++j)
#pragma unroll
          for (int n = 0; n < NTW; ++n) {
            if (j == 0) {
              f32x4_t z = f32x4_t{0.f, 0.f, 0.f, 0.f};
              mma_chunk<BF16>(z, W[0][j][n], xf[0][j]);
              acc[S0][i][n] = z;
            } else {
              mma_chunk<BF16>(acc[S0][i][n], W[0][j][n], xf[0][j]);
            }
            mma_chunk<BF16>(acc[S1][i][n], W[1][j][n], xf[0][j]);
            mma_chunk<BF16>(acc[S2][i][n], W[2][j][n], xf[0][j]);
          }
      }
      if (store) store_tile(std::integral_constant<int, SD>{}, i, orsrc);
    }
  };
  // after the last plane: the set it completed (output index pi - 2 at "iteration" pi = nin) still has to leave
  auto drain = [&](int pi, auto rc) __attribute__((always_inline)) {
    constexpr int SD = (decltype(rc)::value + 2) % 4;
    if (CT3_ABL(4)) return;
    const __amdgpu_buffer_rsrc_t orsrc = out_rsrc(pi - 2);
#pragma unroll
    for (int i = 0; i < MT; ++i) store_tile(std::integral_constant<int, SD>{}, i, orsrc);
  };
  using std::integral_constant;
  {
    int pi = 0;
    for (; pi + 4 <= nin; pi += 4) {
      plane(pi, integral_constant<int, 0>{});
      plane(pi + 1, integral_constant<int, 1>{});
      plane(pi + 2, integral_constant<int, 2>{});
      plane(pi + 3, integral_constant<int, 3>{});
    }
    if (pi >= nin) {
      drain(pi, integral_constant<int, 0>{});
    } else {
      plane(pi, integral_constant<int, 0>{});
      if (pi + 1 >= nin) {
        drain(pi + 1, integral_constant<int, 1>{});
      } else {
        plane(pi + 1, integral_constant<int, 1>{});
        if (pi + 2 >= nin) {
          drain(pi + 2, integral_constant<int, 2>{});
        } else {
          plane(pi + 2, integral_constant<int, 2>{});
          drain(pi + 3, integral_constant<int, 3>{});
        }
      }
    }
  }
  // (DMAs issued past the last plane carry zero-record descriptors and land in buffers nobody reads again; they are
  // drained before the wave ends so that the workgroup's LDS is not handed on with writes in flight)
  ct3_wait<0>();
}

template <int CP, int WN, int NTW, int MT, int TZ, int WAVES, bool PS>
int launch_thin3_tz(Ct3Args& a, hipStream_t st) {
  using G = Ct3Geom<CP, WN, NTW, MT, TZ, WAVES, PS>;
  a.nty = (a.Y + G::TY - 1) / G::TY;
  a.ntz = (a.Z + TZ - 1) / TZ;
  // x segments: enough workgroups for two per CU; a segment re-reads two halo planes
  const int cols = a.B * a.nty * a.ntz;
  int nseg = (WSR_ENV_INT("WSR_CT3_WGS", 4096 / WAVES) + cols - 1) / cols;
  if (nseg < 1) nseg = 1;
  int xs = (a.X + nseg - 1) / nseg;
  if (xs < 8) xs = a.X < 8 ? a.X : 8;
  a.XS = xs;
  a.nseg = (a.X + xs - 1) / xs;
  if ((long)a.Y * a.Z * a.in_ctot * 2 >= (long)CT3_OOB) return WSR_EUNSUPPORTED;  // plane offsets are 32-bit
  if ((long)a.Y * a.Z * a.out_ctot * 2 >= (long)CT3_OOB) return WSR_EUNSUPPORTED;
  auto kern = conv_thin3_kernel<CP, WN, NTW, MT, TZ, WAVES, PS>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
#ifdef WSR_CT3_ABL_RT
  a.ablate = WSR_ENV_INT("WSR_CT3_ABL", 0);
#endif
  const size_t lds = (size_t)G::NB * G::STRIDE;
  hipLaunchKernelGGL(kern, dim3((unsigned)(cols * a.nseg)), dim3(WAVES * 64), lds, st, a);
  WSR_LAUNCH_CHECK();
  return 0;
}

// column tile: 32 levels when the volume's levels allow (fewer halo columns per voxel), else 16
template <int CP, int WN, int NTW, int MT, int WAVES = 8>
int launch_thin3(Ct3Args& a, hipStream_t st) {
  if (a.Z % 16) return WSR_EUNSUPPORTED;
  // one n-tile per wave, an even number of m-tiles, whole 16-byte pieces of the output window: paired stores (Ct3Geom)
  constexpr bool CAN_PAIR = NTW == 1 && MT % 2 == 0;
  // Measured (round 5, same-device A/B at C3-literal, where these launches last 0.2-0.6 ms): the one-n-tile-per-workgroup convs
  // do NOT gain - terrain 1 -> 16: 315 -> 322 us, 16 -> 16: 555 -> 581 / 528 -> 540 - they are not store-issue-bound (the
  // guide's T21 test: halve the store instructions at equal bytes and nothing moves); the 4 -> 128 feature conv, whose eight
  // waves each write a 32-byte slice of every 256-byte voxel row, does: 194 -> 182 us.  So: paired stores where several
  // waves share a voxel row (WN > 1); WSR_CT3_PAIR=1 / 0 forces them on / off everywhere.
  const int want = WSR_ENV_INT("WSR_CT3_PAIR", WN > 1 ? 1 : 0);
  const bool ps = CAN_PAIR && want && a.N % 8 == 0 && a.out_ctot % 8 == 0 && a.out_off % 8 == 0;
  if constexpr ((WAVES / WN) * MT >= 8 && ((WAVES / WN) * MT) % 2 == 0) {
    if (a.Z % 32 == 0 && !WSR_ENV_SET("WSR_CT3_TZ16")) {
      if constexpr (CAN_PAIR) {
        if (ps) return launch_thin3_tz<CP, WN, NTW, MT, 32, WAVES, true>(a, st);
      }
      return launch_thin3_tz<CP, WN, NTW, MT, 32, WAVES, false>(a, st);
    }
  }
  if constexpr (CAN_PAIR) {
    if (ps) return launch_thin3_tz<CP, WN, NTW, MT, 16, WAVES, true>(a, st);
  }
  return launch_thin3_tz<CP, WN, NTW, MT, 16, WAVES, false>(a, st);
}

}  // namespace

// 3x3x3, stride 1, padding 1 conv with red <= 16 stored reduction channels on the sliding-window kernel.  `wfrag`: the
// tile kernels' fragment filter of this conv (wsr_pack_filter_frag; for an input gradient the transposed, tap-flipped
// one).  WSR_EUNSUPPORTED: shape outside the instantiations - the caller goes on to the halo-tile kernel.
int wsr_conv_thin3(const unsigned short* in, int in_ctot, int in_off, int red, const unsigned short* wfrag,
                   unsigned short* out, int out_ctot, int out_off, int n_out, int B, int X, int Y, int Z, const float* bias,
                   float alpha, int act, float slope, hipStream_t st) {
  if (WSR_ENV_SET("WSR_NO_THIN")) return WSR_EUNSUPPORTED;  // tuning / A-B switch
  if (red != 8 && red != 16) return WSR_EUNSUPPORTED;
  if (in_ctot % 8 || in_off % 8 || out_ctot % 4 || out_off % 4 || n_out % 4) return WSR_EUNSUPPORTED;
  if (Z % 16 || Z < 16 || Y < 4 || X < 4) return WSR_EUNSUPPORTED;
  if ((long)B * X * Y * Z < 4096) return WSR_EUNSUPPORTED;  // tiny volumes: latency-bound either way
  if (bias && ((size_t)bias & 3)) return WSR_EUNSUPPORTED;
  Ct3Args a{};
  a.in = in; a.wf = wfrag; a.out = out; a.bias = bias;
  a.alpha = alpha; a.slope = slope; a.act = act;
  a.B = B; a.X = X; a.Y = Y; a.Z = Z;
  a.in_ctot = in_ctot; a.in_off = in_off; a.out_ctot = out_ctot; a.out_off = out_off;
  a.N = n_out; a.NT_total = (n_out + 15) / 16;
  const int nt = a.NT_total;
  const int w4 = WSR_ENV_INT("WSR_CT3_W4", 0);  // tuning: four-wave workgroups (more, smaller barrier domains per CU)
  if (w4) {
    if (red == 16 && nt == 1) return w4 == 1 ? launch_thin3<16, 1, 1, 2, 4>(a, st) : launch_thin3<16, 1, 1, 4, 4>(a, st);
    if (red == 8 && nt == 1) return w4 == 1 ? launch_thin3<8, 1, 1, 2, 4>(a, st) : launch_thin3<8, 1, 1, 4, 4>(a, st);
    if (red == 8 && nt == 2 && n_out % 8 == 0 && out_ctot % 8 == 0 && out_off % 8 == 0)
      return w4 == 1 ? launch_thin3<8, 1, 2, 1, 4>(a, st) : launch_thin3<8, 1, 2, 2, 4>(a, st);
  }
  if (red == 16) {
    if (nt == 1) return launch_thin3<16, 1, 1, 2>(a, st);
    if (nt == 2 && n_out % 8 == 0 && out_ctot % 8 == 0 && out_off % 8 == 0) return launch_thin3<16, 1, 2, 1>(a, st);
    return WSR_EUNSUPPORTED;
  }
  if (nt == 1) return launch_thin3<8, 1, 1, 2>(a, st);
  // (32 outputs from 8 stored channels - the discriminator's first conv - is all stores: four-wave workgroups with two
  // m-tiles per wave measured 96 us against 116 for eight waves with one, on 2 x 128^3 voxels)
  if (nt == 2 && n_out % 8 == 0 && out_ctot % 8 == 0 && out_off % 8 == 0)
    return WSR_ENV_SET("WSR_CT3_D8") ? launch_thin3<8, 1, 2, 1>(a, st) : launch_thin3<8, 1, 2, 2, 4>(a, st);
  if (nt == 2) return WSR_EUNSUPPORTED;
  if (nt <= 8) return launch_thin3<8, 8, 1, 4>(a, st);
  return WSR_EUNSUPPORTED;
}
