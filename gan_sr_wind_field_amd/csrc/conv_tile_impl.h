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
//
// Element type T (last template parameter): BF16 (above) or F32 - the reference's own arithmetic (AMP is commented
// out there, Generator_3D_Resnet_ESRGAN.py:65).  Everything is laid out in 16-byte PIECES (8 bf16 or 4 fp32 channels
// of one voxel), so the fp32 kernel is the same program on half as many channels per chunk (CK = 16 / TPK instead of
// 32 / TPK): identical LDS images, DMA units and 1 KB weight fragments; one K-step is four v_mfma_f32_16x16x4_f32
// (exact fp32 products and sums) instead of one v_mfma_f32_16x16x32_bf16; epilogue operands are 16 instead of 8 bytes.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.h"

struct CtArgs {
  const unsigned short* in;
  const unsigned short* wf;  // fragment-packed filter: [chunk][kstep][ntile][64 lanes][8]
  void* out;
  const void* zero16;        // 16 zero bytes in global memory (source of out-of-range DMA lanes)
  const float* bias;
  const float* chan_scale;
  const unsigned short* res;
  int res_ctot, res_off, res_c1;  // residual on produced channels < res_c1 only
  float alpha, beta, slope;
  int act, out_planar;            // act: 0 none, 1 LeakyReLU, 2 LeakyReLU after the residual add
  int act_c1;                     // bias + activation apply to channels < act_c1 (others: raw sums)
  int B, Xi, Yi, Zi, Xo, Yo, Zo, ups;
  int in_ctot, in_off, nchunks;   // reduction channels = nchunks * CK, window [in_off, ...)
  int cin_valid;                  // channels of the window that exist in memory (multiple of 8)
  int Cout, out_ctot, out_off;
  int KX, KY, KZ, px, py, pz;
  int sx, sy, sz;   // output stride of the gather (forward convs of the discriminator's down-sampling layers; 1 elsewhere)
  int TX, TY, TZ;
  int tiles_x, tiles_y, tiles_z, ntiles;
  int nts;          // K-steps per chunk = ceil(taps / TPK)
  int TS;           // K-steps per weight stage
  int NT_total;     // 16-wide output-channel tiles
  int ngroups;      // n-tile groups (grid = ntiles * ngroups)
  int P;            // activation plane stride (bytes)
  int off_mtab, off_htab, off_ttab, off_btab, off_xs, off_ws;  // btab: bias and scale of the workgroup's columns
  int vec_ok;
  const unsigned short* mask_y;  // LeakyReLU-backward mask source (saved forward output) or NULL
  int mask_ctot, mask_off, mask_c0, mask_c1;
  float mask_slope;
  int xbufs;        // activation buffers in LDS: 2 = next chunk prefetched during the MFMAs
  // ceil(2^32 / d) for the runtime divisors of the prologue (fdiv): tile / halo / tap extents, tile counts
  unsigned mg_TZ, mg_TY, mg_Lz, mg_Ly, mg_KZ, mg_KY, mg_ng, mg_tz, mg_ty, mg_tx;
  // Sub-pixel form of an up-sampling conv (wsr_conv_t.lat): the gathered tensor (il_*) or the produced tensor (ol_*)
  // is the sub-lattice (m*x + ox, m*y + oy, z) of a tensor m times as large along x and y (m = 1: the tensor itself).
  // nphase = 4: the four parity convs in one launch - parity (a, b) = bits of the tile index: pads (px - a, py - b),
  // produced lattice offsets (a, b), filter wf + (2a + b) * ph_wstride.
  int il_m, il_ox, il_oy, ol_m, ol_ox, ol_oy, nphase;
  int ol_mz, ol_oz;  // the same for the produced tensor's z axis (input gradients of z-strided convs)
  long ph_wstride;
  // Split reduction (launches with few workgroups and long reductions: the deep layers of the discriminator): the
  // grid is ksplit times as large, split s contracts chunks [s*cps, (s+1)*cps) and stores its raw fp32 sums to
  // part + s*part_stride as [voxel][16*NT_total] rows; splitk_reduce_kernel adds the splits in order and applies the
  // epilogue.  ws / ws_bytes: the caller's workspace (wsr_conv_tile_workspace), NULL = never split.
  int ksplit, cps;
  float* part;
  long part_stride;
  void* ws;
  long ws_bytes;
  // Single activation buffer (the 5x5x5 144 -> 144 conv: no room for two): the halo image of the NEXT chunk is loaded
  // in two parts, both under MFMAs.  The K-steps walk the taps kx-major, so the last tap column (kx = KX-1) only reads
  // x-planes >= KX-1 of the image: from weight stage xs_stage on, planes < KX-1 (DMA units [0, xs_units)) are dead and
  // take the next chunk; the rest is issued when the next chunk starts, whose first stage (kx = 0 only) reads planes
  // < TX.  xs_stage < 0: off (the whole image is reloaded between chunks behind an exposed wait + barrier).
  int xs_stage, xs_units;
  // The gathered tensor in TWO parts (wsr_epilogue_t.in2: the generator's concat in front of the 5x5x5 conv): reduction
  // chunks >= in2_chunk come from channels [0, ...) of in2 (in2_ctot channels per voxel).  And its mirror image for the
  // input gradient (wsr_dgrad_opts_t.dx2): produced channels >= out2_c0 go to channels [0, ...) of out2.  TN = 9
  // instantiations only (ct_two_src): the others never look at these fields.
  const unsigned short* in2;
  int in2_ctot, in2_chunk;
  void* out2;
  int out2_ctot, out2_c0;
  int f32;     // 1: fp32 operands (the element type of in / wf / res / mask_y / out; see the kernel's T)
  int prio;    // 1: waves of the second half of the workgroup run the main loop at s_setprio 1 (tuning switch)
  int ablate;  // -DWSR_CT_STAMPS builds, timing only: skip 1 = activation prefetch, 2 = weight prefetch, 4 = LDS reads, 8 = MFMAs; 32 = paired 16-byte epilogue stores
  unsigned long long* stamps;  // -DWSR_CT_STAMPS builds: [workgroup][8] clock samples of wave 0 (else unused)
};

// adds the ksplit partial images of a split-reduction launch in index order and applies the epilogue (conv_tile.hip)
int wsr_ct_splitk_reduce(const CtArgs& a, hipStream_t st);

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

// In-kernel phase stamps (tuning builds only): wave 0 of every workgroup samples the 100 MHz wall clock
// (slots 0-5) and the shader clock at entry / exit (slots 6-7).
#ifdef WSR_CT_STAMPS
#define CT_STAMP(k)                                                                                   \
  do {                                                                                                \
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + (k)] = (k) >= 6 ? clock64() : wall_clock64(); \
  } while (0)
#else
#define CT_STAMP(k) do {} while (0)
#endif

// n / d for n < 65536, d < 65536 with mg = ceil(2^32 / d) (exact: the error term n*(mg*d - 2^32) stays
// below 2^32); d == 1 has no 32-bit multiplier and is passed through.
__device__ __forceinline__ unsigned fdiv(unsigned n, unsigned d, unsigned mg) { return d == 1 ? n : __umulhi(n, mg); }
static inline unsigned fdiv_magic(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); }

// instantiations that take the gathered / produced tensor in two parts (CtArgs.in2 / out2): the 512-voxel 9-n-tile
// workgroups of the 5x5x5 144 -> 144 conv - everywhere else the extra address arithmetic would be paid for nothing
constexpr bool ct_two_src(int wn, int tn) { return wn == 1 && tn == 9; }
// activation DMA units (1 KB) a wave may have to issue per chunk: registers of the resolved source offsets
constexpr int ct_xk(int waves, int tm) { return waves <= 4 ? 16 : (tm <= 2 ? 13 : 10); }

// (four-wave workgroups: one wave per SIMD by construction - let the register allocator have all 512)
// 4 consecutive channels as stored: 8 bytes of bf16, 16 of fp32 (epilogue operands)
template <class T> struct ct_v4 { using type = uint2; };
template <> struct ct_v4<F32> { using type = uint4; };
template <class T> __device__ __forceinline__ float4 ct_cvt4(const typename ct_v4<T>::type& v);
template <> __device__ __forceinline__ float4 ct_cvt4<BF16>(const uint2& v) {
  return make_float4(bf2f((unsigned short)(v.x & 0xFFFFu)), bf2f((unsigned short)(v.x >> 16)),
                     bf2f((unsigned short)(v.y & 0xFFFFu)), bf2f((unsigned short)(v.y >> 16)));
}
template <> __device__ __forceinline__ float4 ct_cvt4<F32>(const uint4& v) {
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// (y > 0 ? 1 : slope) per element, from the stored bits (bf16: sign test on the raw halves)
template <class T> __device__ __forceinline__ float4 ct_mask4(const typename ct_v4<T>::type& y, float slope);
template <> __device__ __forceinline__ float4 ct_mask4<BF16>(const uint2& y, float slope) {
  return make_float4((short)(y.x & 0xFFFFu) > 0 ? 1.f : slope, (int)y.x > 0xFFFF ? 1.f : slope,
                     (short)(y.y & 0xFFFFu) > 0 ? 1.f : slope, (int)y.y > 0xFFFF ? 1.f : slope);
}
template <> __device__ __forceinline__ float4 ct_mask4<F32>(const uint4& y, float slope) {
  return make_float4(__uint_as_float(y.x) > 0.f ? 1.f : slope, __uint_as_float(y.y) > 0.f ? 1.f : slope,
                     __uint_as_float(y.z) > 0.f ? 1.f : slope, __uint_as_float(y.w) > 0.f ? 1.f : slope);
}
template <class T> __device__ __forceinline__ typename ct_v4<T>::type ct_ones4();
template <> __device__ __forceinline__ uint2 ct_ones4<BF16>() { return make_uint2(0x3F803F80u, 0x3F803F80u); }
template <> __device__ __forceinline__ uint4 ct_ones4<F32>() {
  return make_uint4(0x3F800000u, 0x3F800000u, 0x3F800000u, 0x3F800000u);
}
template <class T> __device__ __forceinline__ typename ct_v4<T>::type ct_zero4();
template <> __device__ __forceinline__ uint2 ct_zero4<BF16>() { return make_uint2(0u, 0u); }
template <> __device__ __forceinline__ uint4 ct_zero4<F32>() { return make_uint4(0u, 0u, 0u, 0u); }

// WK > 1 (small volumes, conv_tile_small.hip): WK waves share each (m-tile group, n-tile group) and split the K-STEPS of
// every weight stage between them - a volume of 80 tiles leaves three quarters of the chip's SIMDs without a wave while
// each of its two-wave workgroups walks 28..162 dependent K-steps; with the reduction over four waves the same workgroup
// is four times shorter, and the partial sums meet in LDS (fixed order: bit-reproducible) before the epilogue.
//
// SIMPLE != 0 (conv_tile_simple_*.hip: the trunk's launches - stride 1, no lattice / parity / up-sampling gather, no split
// reduction, no planar output, whole 4-channel groups; launch_ct checks the list): the general forms' run-time switches
// become constants.  The kernel takes ~100 uniform arguments and the general epilogue tests many of them per (m-tile,
// n-tile): the 32-wide instantiation carried 108-185 spilled SGPRs (2 559 v_readlane in 9 500 lines of ISA, a third of its
// epilogue), i.e. a prologue and an epilogue paid in SGPR reloads on launches that last 16-30 us.  SIMPLE == 1 (every shipped
// instantiation): also no per-sample channel scale and no two-tensor concat; == 2 keeps those two (the 144-wide tile of the
// 5x5x5 conv: measured, no gain - its 7 ms are main loop - and not instantiated in the library).
template <int WM, int WN, int TM, int TN, int TPK, bool PIPE, bool MASK, class T = BF16, int WK = 1, int SIMPLE = 0>
__global__ __launch_bounds__(WM * WN * WK * 64) __attribute__((amdgpu_waves_per_eu(1, WM * WN * WK <= 4 ? 1 : 8)))
void conv_tile_kernel(const CtArgs a) {
  using E = typename T::elem;
  using V4 = typename ct_v4<T>::type;
  constexpr int EPP = T::EPP;      // channels per 16-byte piece
  constexpr int WAVES = WM * WN * WK, NT = WAVES * 64;
  constexpr int PL = 4 / TPK;      // pieces (bf16: channel octets) per chunk and voxel
  constexpr int CK = EPP * PL;     // channels per chunk
  constexpr int NTW = WN * TN;     // n-tiles per workgroup
  constexpr bool TWO = ct_two_src(WN, TN);  // this instantiation reads a two-tensor concat / writes a two-tensor gradient
  constexpr bool STAGGER = true;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wk = WK > 1 ? wave / (WM * WN) : 0;            // which share of every stage's K-steps (waves 0 .. WM*WN-1: share 0)
  const int wmn = WK > 1 ? wave - wk * (WM * WN) : wave;
  const int wm = wmn / WN, wn = wmn % WN;
  // (SIMPLE: constants; else the arguments)
  const int sx = SIMPLE ? 1 : a.sx, sy = SIMPLE ? 1 : a.sy, sz = SIMPLE ? 1 : a.sz;
  const int il_m = SIMPLE ? 1 : a.il_m, il_ox = SIMPLE ? 0 : a.il_ox, il_oy = SIMPLE ? 0 : a.il_oy;
  const int ol_m = SIMPLE ? 1 : a.ol_m, ol_mz = SIMPLE ? 1 : a.ol_mz;
  const int ol_ox = SIMPLE ? 0 : a.ol_ox, ol_oy = SIMPLE ? 0 : a.ol_oy, ol_oz = SIMPLE ? 0 : a.ol_oz;
  const int nphase = SIMPLE ? 1 : a.nphase, ups = SIMPLE ? 0 : a.ups, ksplit = SIMPLE ? 1 : a.ksplit;
  const bool out_planar = SIMPLE ? false : (a.out_planar != 0);
  const float* const chan_scale = SIMPLE == 1 ? nullptr : a.chan_scale;
  CT_STAMP(0);
  CT_STAMP(6);

  const int Lx = (a.TX - 1) * sx + a.KX, Ly = (a.TY - 1) * sy + a.KY, Lz = (a.TZ - 1) * sz + a.KZ;
  const int L = Lx * Ly * Lz;
  const int M = a.TX * a.TY * a.TZ;  // <= MR
  constexpr int MR = WM * TM * 16;    // MFMA rows of the workgroup
  const int taps = a.KX * a.KY * a.KZ;

  unsigned* mtab = reinterpret_cast<unsigned*>(smem + a.off_mtab);
  unsigned short* htab = reinterpret_cast<unsigned short*>(smem + a.off_htab);
  int* ttab = reinterpret_cast<int*>(smem + a.off_ttab);
  char* Xs = smem + a.off_xs;
  char* Ws = smem + a.off_ws;

  unsigned bid = (unsigned)xcd_remap(blockIdx.x, gridDim.x);
  int c_begin = 0, nchunks_l = a.nchunks;  // this workgroup's slice of the reduction channels
  float* part = nullptr;
  if (ksplit > 1) {
    const unsigned per = gridDim.x / (unsigned)ksplit, ksi = bid / per;
    bid -= ksi * per;
    c_begin = (int)ksi * a.cps;
    nchunks_l = min(a.cps, a.nchunks - c_begin);
    part = a.part + (size_t)ksi * a.part_stride;
  }
  const unsigned tile = fdiv(bid, a.ngroups, a.mg_ng);
  const int ng = (int)(bid - tile * a.ngroups);
  // the four parity convs of a sub-pixel launch sit side by side in the grid: they gather the same halo (L2)
  const int pha = nphase == 4 ? (int)((tile >> 1) & 1) : 0, phb = nphase == 4 ? (int)(tile & 1) : 0;
  const int ppx = a.px - pha, ppy = a.py - phb;
  unsigned r = nphase == 4 ? tile >> 2 : tile, r2;
  r2 = fdiv(r, a.tiles_z, a.mg_tz); const int tz = (int)(r - r2 * a.tiles_z); r = r2;
  r2 = fdiv(r, a.tiles_y, a.mg_ty); const int ty = (int)(r - r2 * a.tiles_y); r = r2;
  r2 = fdiv(r, a.tiles_x, a.mg_tx); const int tx = (int)(r - r2 * a.tiles_x);
  const int b = (int)r2;
  const int x0 = tx * a.TX, y0 = ty * a.TY, z0 = tz * a.TZ;
  const int nt0 = ng * NTW;  // first n-tile of this workgroup

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

  const int U = ups ? 1 : 0;
  const int nstages = (a.nts + a.TS - 1) / a.TS;
  const int stage_units = a.TS * NTW;  // 1 KB fragments per weight stage
  const int UPP = VM ? (L + 31) >> 5 : (L + 63) >> 6;  // 1 KB DMA units per activation plane (VM: per chunk)
  const int HU = VM ? UPP : UPP * PL;                   // ... per chunk
  const int xs_bytes = VM ? a.P : PL * a.P;
  const char* wbase = reinterpret_cast<const char*>(a.wf) + (size_t)(2 * pha + phb) * a.ph_wstride * sizeof(E) +
                      (size_t)nt0 * 1024;  // (fragments are 1 KB whatever the element type)
  typedef __attribute__((address_space(3))) char* lptr_t;
  const unsigned xs_lds = (unsigned)(unsigned long)(lptr_t)Xs;  // LDS byte addresses
  const unsigned ws_lds = (unsigned)(unsigned long)(lptr_t)Ws;

  // LDS-DMA (global_load_lds): each wave moves whole 1 KB units, lane l -> unit base + 16*l.
  // weights of stage (chunk, st) -> Ws[buf]
  auto w_issue = [&](int chunk, int st, int buf) __attribute__((always_inline)) {
    const unsigned dst = ws_lds + buf * stage_units * 1024;
    for (int u = wave; u < stage_units; u += WAVES) {
      const int tsi = u / NTW, nl = u - tsi * NTW;
      const int ts = st * a.TS + tsi;
      if (ts < a.nts && nt0 + nl < a.NT_total) {
        const char* src = wbase + ((size_t)((chunk + c_begin) * a.nts + ts) * a.NT_total + nl) * 1024 + lane * 16;
        glds16(src, __builtin_amdgcn_readfirstlane(dst + u * 1024));
      }
    }
  };
  // The first weight stage needs nothing computed here: its DMA flies while the tables are built.
  w_issue(0, 0, 0);

  // The halo geometry is the same for every chunk, so each wave resolves the source of "its" DMA units
  // (u = wave + WAVES*k) once: element offset of the lane's voxel (or OOB), octet plane, LDS offset.
  constexpr int XK = ct_xk(WAVES, TM);  // max units per wave per chunk (checked on the host; strided tiles: 13)
  // Offsets are 32-bit and RELATIVE to the first x-plane the tile's halo touches (a 64-bit workgroup-uniform base):
  // tensors beyond 2^32 elements (the literal 128^3 -> 512 x 512 x 128 reading of BASELINE.json configs[2]: 4.8e9 in a
  // 144-channel HR tensor) only need the halo's few planes to stay below 2^32 elements (checked on the host).
  const int gx_lo = max(x0 * sx - ppx, 0) >> U;  // first stored x-plane of the halo
  const long vox_base = ((long)b * a.Xi + gx_lo) * il_m * ((long)a.Yi * il_m) * a.Zi;
  const E* in_base = reinterpret_cast<const E*>(a.in) + vox_base * a.in_ctot;
  const E* in2_base = TWO && a.in2 ? reinterpret_cast<const E*>(a.in2) + vox_base * a.in2_ctot : nullptr;
  unsigned xoff[XK];  // element offset of the lane's voxel + window + piece (TWO: the voxel index; the rest per issue)
  int xo8[VM ? 1 : XK], xdst[VM ? 1 : XK];  // voxel-major rows: octet = lane & 1, unit u lands at u KB
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
      } else {
        pl = u / UPP;
        v = (u - pl * UPP) * 64 + lane;
        dsto = pl * a.P + (u - pl * UPP) * 1024;
      }
      if (v < L) {
        const unsigned q = fdiv((unsigned)v, Lz, a.mg_Lz), hx = fdiv(q, Ly, a.mg_Ly);
        const int gx = x0 * sx - ppx + (int)hx, gy = y0 * sy - ppy + (int)(q - hx * Ly),
                  gz = z0 * sz - a.pz + (int)(v - q * Lz);
        if ((unsigned)gx < (unsigned)(a.Xi << U) && (unsigned)gy < (unsigned)(a.Yi << U) &&
            (unsigned)gz < (unsigned)a.Zi) {
          // 32-bit arithmetic on the voxel index relative to plane gx_lo (the host checked the halo's extent)
          const unsigned vox = ((((unsigned)((gx >> U) - gx_lo)) * il_m + il_ox) * (a.Yi * il_m) +
                                (gy >> U) * il_m + il_oy) * a.Zi + gz;
          off = TWO ? vox : vox * (unsigned)a.in_ctot + (unsigned)(a.in_off + EPP * pl);
        }
      }
    }
    xoff[k] = off;
    if constexpr (!VM) {
      xo8[k] = EPP * pl;
      xdst[k] = __builtin_amdgcn_readfirstlane(dsto);
    }
  }
  if constexpr (VM) { xo8[0] = EPP * (lane & 1); xdst[0] = 0; }
  // units [u0, u1) of the activation chunk (with halo) -> Xs[buf]; out-of-range voxels read the zero page
  auto x_issue = [&](int chunk, int buf, int u0, int u1) __attribute__((always_inline)) {
    const unsigned dst = xs_lds + buf * xs_bytes;
#pragma unroll
    for (int k = 0; k < XK; ++k) {
      const int u = wave + WAVES * k;
      if (u >= u0 && u < u1) {
        const bool ok = xoff[k] != 0xFFFFFFFFu && (chunk + c_begin) * CK + xo8[VM ? 0 : k] < a.cin_valid;
        const E* src;
        if constexpr (TWO) {  // (uniform choice of the tensor per chunk; the offsets are rebuilt from the voxel index)
          const int cg = chunk + c_begin;
          const bool second = in2_base != nullptr && cg >= a.in2_chunk;
          const E* bp = second ? in2_base : in_base;
          const unsigned ct = second ? (unsigned)a.in2_ctot : (unsigned)a.in_ctot;
          const unsigned co = second ? (unsigned)((cg - a.in2_chunk) * CK) : (unsigned)(a.in_off + cg * CK);
          src = ok ? bp + (size_t)(xoff[k] * ct + co + (unsigned)xo8[VM ? 0 : k]) : reinterpret_cast<const E*>(a.zero16);
        } else {
          src = ok ? in_base + (size_t)xoff[k] + (chunk + c_begin) * CK : reinterpret_cast<const E*>(a.zero16);
        }
        glds16(src, dst + (VM ? u * 1024 : xdst[VM ? 0 : k]));
      }
    }
  };
  CT_STAMP(1);
  x_issue(0, 0, 0, HU);
  CT_STAMP(2);

  // ---- tables (built under the first DMA) ---------------------------------------------------
  for (int m = t; m < MR; m += NT) {  // rows >= M (tile volume) are padding: flagged, read voxel 0
    const unsigned q = fdiv((unsigned)m, a.TZ, a.mg_TZ), ox = fdiv(q, a.TY, a.mg_TY);
    const unsigned oz = m - q * a.TZ, oy = q - ox * a.TY;
    mtab[m] = m < M ? (ox | (oy << 8) | (oz << 16)) : (1u << 24);
    htab[m] = m < M ? (unsigned short)((ox * sx * Ly + oy * sy) * Lz + oz * sz) : (unsigned short)0;
  }
  // per-channel epilogue constants of this workgroup's columns: bias (where it applies) and
  // channel scale * alpha - fetched now, so that the epilogue finds them in LDS instead of waiting for
  // global memory once per n-tile
  float* btab = reinterpret_cast<float*>(smem + a.off_btab);
  for (int k = t; k < NTW * 16; k += NT) {
    const int co = nt0 * 16 + k;
    const bool in = co < a.Cout;
    btab[k] = (a.bias && in && co < a.act_c1) ? a.bias[co] : 0.f;
    btab[NTW * 16 + k] = ((chan_scale && in) ? chan_scale[(long)b * a.Cout + co] : 1.f) * a.alpha;
  }
  for (int k = t; k < a.nts * TPK; k += NT) {
    int off = 0;
    if (k < taps) {
      const unsigned q = fdiv((unsigned)k, a.KZ, a.mg_KZ), kx = fdiv(q, a.KY, a.mg_KY);
      off = (int)((kx * Ly + (q - kx * a.KY)) * Lz + (k - q * a.KZ));
    }
    ttab[k] = off;
  }
  dma_wait();
  __syncthreads();
  CT_STAMP(3);

  int hb[TM];                              // byte offset of row `fr` of m-tile i in a plane
#pragma unroll
  for (int i = 0; i < TM; ++i) hb[i] = (int)htab[(wm * TM + i) * 16 + fr] * RB + lane_plane;

  f32x4_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#ifdef WSR_CT_STAMPS
  long long st_dma = 0, st_bar = 0;  // shader cycles this wave spent waiting for its DMAs / at the phase barrier
#endif
  // Epilogue operands requested EARLY (round 6; the 32-wide SIMPLE instantiations: one workgroup per CU by LDS, so the
  // registers are free): the partial sums / running gradient a launch adds to (`res`) and the saved activation its
  // LeakyReLU mask is taken from were written by earlier launches - their loads go out at the top of the LAST phase and
  // land under its K-steps instead of costing a global-memory round trip between the main loop and the first store.
  // Measured and NOT shipped (build the 32-wide translation units with -DWSR_CT_EPF to have it; profiles/r06_i_ab_epilogue_prefetch.txt):
  // C3' 90.12 / 90.17 ms with, 89.93 / 90.07 without - the loads' latency was not what the 3 us epilogue is made of - and the
  // 145-164 registers it takes end the two-workgroups-per-CU form of multi-round launches (C4 332.2 vs 326.8 ms).
#ifdef WSR_CT_EPF
  constexpr bool EPF = SIMPLE == 1 && TN <= 2 && WK == 1;
#else
  constexpr bool EPF = false;
#endif
  V4 pre_rr[EPF ? TN : 1][EPF ? TM : 1], pre_yy[EPF ? TN : 1][EPF ? TM : 1];
  auto epilogue_prefetch = [&]() __attribute__((always_inline)) {
    if constexpr (EPF) {
      const int cob_ = (nt0 + wn * TN) * 16 + fg * 4;
      const long vpb = (long)a.Xo * a.Yo * a.Zo;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const unsigned mv = mtab[(wm * TM + i) * 16 + fr];
        const int gx = x0 + (int)(mv & 255), gy = y0 + (int)((mv >> 8) & 255), gz = z0 + (int)((mv >> 16) & 255);
        const bool ok = !(mv >> 24) && gx < a.Xo && gy < a.Yo && gz < a.Zo;
        const long m = ok ? (long)b * vpb + ((long)gx * a.Yo + gy) * a.Zo + gz : -1;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int co0 = cob_ + 16 * j;
          pre_rr[j][i] = ct_zero4<T>();
          pre_yy[j][i] = ct_ones4<T>();
          if (m < 0 || co0 >= a.Cout) continue;
          if (a.res && co0 < a.res_c1)
            pre_rr[j][i] = *reinterpret_cast<const V4*>(reinterpret_cast<const E*>(a.res) + m * a.res_ctot + a.res_off + co0);
          if (MASK && a.mask_y && co0 >= a.mask_c0 && co0 < a.mask_c1)
            pre_yy[j][i] = *reinterpret_cast<const V4*>(reinterpret_cast<const E*>(a.mask_y) + m * a.mask_ctot + a.mask_off +
                                                        (co0 - a.mask_c0));
        }
      }
    }
  };
  const int total_phases = nchunks_l * nstages;
  int chunk = 0, st = 0;
  // static priority for the second-dispatched half (it loses the issue arbitration against the older half on
  // every phase otherwise: MI355X_MICROARCH.md, two waves per SIMD, item 4)
  if (a.prio && wave >= WAVES / 2) __builtin_amdgcn_s_setprio(1);
  for (int ph = 0; ph < total_phases; ++ph) {
    if constexpr (EPF) {
      if (ph + 1 == total_phases) epilogue_prefetch();
    }
    // ---- prefetch: next weight stage, and a slice of the next chunk's activations.  The burst costs each
    // wave several hundred issue cycles during which it feeds no MFMAs, so the two halves of the workgroup
    // (waves w and w + WAVES/2 share a SIMD) take turns: the first half issues at the top of the phase, the
    // second half in the middle of its K-step loop, and the other wave keeps the SIMD's matrix pipe busy.
    if (a.xbufs != 2) {  // single activation buffer
      // (compiled into the 144-wide instantiation only: with the two extra copies of the issue loop in every
      // instantiation the 32-wide launches of the dense blocks went from 26.8 to 30.7 us)
      if (TN == 9 && a.xs_stage >= 0) {  // next chunk's image in two parts, both under MFMAs (see CtArgs.xs_stage)
        if (st == a.xs_stage && chunk + 1 < nchunks_l) x_issue(chunk + 1, 0, 0, a.xs_units);
        if (st == 0 && chunk > 0) x_issue(chunk, 0, a.xs_units, HU);
      } else if (st == 0 && chunk > 0) {  // reload it between chunks
        x_issue(chunk, 0, 0, HU);
        dma_wait();
        __syncthreads();
      }
    }
    auto burst = [&]() __attribute__((always_inline)) {
#ifdef WSR_CT_STAMPS
      if (a.ablate & 2) goto skip_w;
#endif
      if (ph + 1 < total_phases) {
        const bool wrap = st + 1 == nstages;
        w_issue(wrap ? chunk + 1 : chunk, wrap ? 0 : st + 1, (ph + 1) & 1);
      }
#ifdef WSR_CT_STAMPS
    skip_w:
      if (a.ablate & 1) return;
#endif
      if (a.xbufs == 2 && chunk + 1 < nchunks_l)
        x_issue(chunk + 1, (chunk + 1) & 1, (HU * st) / nstages, (HU * (st + 1)) / nstages);
    };

    const char* wcur = Ws + (ph & 1) * stage_units * 1024 + (wn * TN) * 1024 + lane * 16;
    const char* xcur = Xs + (a.xbufs == 2 ? (chunk & 1) * xs_bytes : 0);
    const int ts_end = min(a.TS, a.nts - st * a.TS);
    const int* tt = ttab + st * a.TS * TPK + lane_tsub;
    auto load_frags = [&](int tsi, uint4 (&wf)[TN], uint4 (&xf)[TM]) __attribute__((always_inline)) {
#ifdef WSR_CT_STAMPS
      if (a.ablate & 4) return;
#endif
      const int toff = tt[tsi * TPK] * RB;
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const uint4*>(wcur + (tsi * NTW + j) * 1024);
#pragma unroll
      for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const uint4*>(xcur + hb[i] + toff);
    };
    auto mma_frags = [&](const uint4 (&wf)[TN], const uint4 (&xf)[TM]) __attribute__((always_inline)) {
#ifdef WSR_CT_STAMPS
      if (a.ablate & 8) return;
#endif
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_chunk<T>(acc[i][j], wf[j], xf[i]);
    };
    auto run_ksteps = [&](int lo, int hi) __attribute__((always_inline)) {
      if (lo >= hi) return;
      if constexpr (WK > 1) {  // this wave's share of the stage: K-steps wk, wk + WK, ...
        for (int tsi = lo + ((wk - lo % WK) + WK) % WK; tsi < hi; tsi += WK) {
          uint4 wf[TN], xf[TM];
          load_frags(tsi, wf, xf);
          mma_frags(wf, xf);
        }
      } else if constexpr (PIPE) {
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
#if defined(WSR_CT_XAHEAD)
        // wide tiles (no room for a second fragment set; the translation unit defines WSR_CT_XAHEAD): the activation
        // fragment of m-tile i+1 is requested before the MFMAs of m-tile i - two in registers instead of the
        // compiler's one, which it fetches AFTER issuing the previous tile's MFMAs and then waits for (ISA: "R wait
        // 9 x MFMA" four times per K-step); WSR_CT_XAHEAD == 2: and the tap offset one K-step ahead.  Measured on the
        // 5x5x5 144 -> 144 conv: 7.68 -> 7.46 ms (= 2), 128-wide tiles -2 % (= 1; = 2 is no better there).  NOT kept:
        // n-tiles outer / m-tiles inner with a weight-fragment ring and the next K-step's activation fragments
        // requested one per n-tile (no LDS round trip exposed at all): 1-3 % SLOWER than this form; reloading the
        // weight fragments in place behind their last use: slower as well.
        int toff = tt[lo * TPK] * RB;
        for (int tsi = lo; tsi < hi; ++tsi) {
          uint4 xf[2], wf[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const uint4*>(wcur + (tsi * NTW + j) * 1024);
#if WSR_CT_XAHEAD == 2
          const int toff_n = tt[(tsi + 1) * TPK] * RB;  // (past the stage's last K-step: read, never used)
#else
          toff = tt[tsi * TPK] * RB;
#endif
          xf[0] = *reinterpret_cast<const uint4*>(xcur + hb[0] + toff);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            if (i + 1 < TM) xf[(i + 1) & 1] = *reinterpret_cast<const uint4*>(xcur + hb[i + 1] + toff);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < TN; ++j) mma_chunk<T>(acc[i][j], wf[j], xf[i & 1]);
            __builtin_amdgcn_sched_barrier(0);
          }
#if WSR_CT_XAHEAD == 2
          toff = toff_n;
#endif
        }
#else
        for (int tsi = lo; tsi < hi; ++tsi) {
          uint4 wf[TN], xf[TM];
          load_frags(tsi, wf, xf);
          mma_frags(wf, xf);
        }
#endif
      }
    };
    const int mid = (STAGGER && wave >= WAVES / 2) ? ts_end / 2 : 0;  // wave-uniform
    // (explicit operand pipelines for the wide tiles - fragment rings, half-K-step prefetch with a fenced
    // scheduler - were measured 8..30 % SLOWER than this plain loop: the compiler's own interleaving of the
    // next K-step's LDS reads with the MFMAs is better than what the register budget leaves room to write)
    if (mid == 0) burst();
    run_ksteps(0, mid);
    if (mid > 0) burst();
    run_ksteps(mid, ts_end);
#ifdef WSR_CT_STAMPS
    const long long tw0 = clock64();
#endif
    dma_wait();  // this wave's prefetches have landed ...
#ifdef WSR_CT_STAMPS
    const long long tw1 = clock64();
#endif
#ifdef WSR_CT_STAMPS
    if (!(a.ablate & 16))  // (timing only, wrong results: what the phase barriers cost - the bound on any barrier-free weight ring)
#endif
    __syncthreads();  // ... and so have everybody else's; the buffers just read are free again
#ifdef WSR_CT_STAMPS
    st_dma += tw1 - tw0;
    st_bar += clock64() - tw1;
#endif
    if (++st == nstages) { st = 0; ++chunk; }
  }

  CT_STAMP(4);
  if constexpr (WK > 1) {
    // the K-step shares meet: waves of share > 0 hand their sums over through LDS (the activation / weight buffers are
    // free: every DMA has landed and the last phase's barrier is behind us), share 0 adds them in share order
    f32x4_t* red = reinterpret_cast<f32x4_t*>(smem + a.off_xs);
    if (wk > 0) {
      f32x4_t* dst = red + (size_t)(((wk - 1) * (WM * WN) + wmn) * TM * TN) * 64 + lane;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) dst[(i * TN + j) * 64] = acc[i][j];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int k = 1; k < WK; ++k) {
      const f32x4_t* src = red + (size_t)(((k - 1) * (WM * WN) + wmn) * TM * TN) * 64 + lane;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const f32x4_t v = src[(i * TN + j) * 64];
          acc[i][j][0] += v[0]; acc[i][j][1] += v[1]; acc[i][j][2] += v[2]; acc[i][j][3] += v[3];
        }
    }
  }
  // ---- epilogue: acc[i][j][r] -> channel (nt0 + wn*TN + j)*16 + 4*fg + r, voxel row fr of m-tile i.
  // No load may sit between two stores (the compiler cannot move it above a store that might alias, so
  // every tile would pay a full memory round trip).  The n-tiles are walked one at a time; the operands
  // of n-tile j+1 (bias, channel scale, residual and mask values of its TM rows) are fetched before the
  // stores of n-tile j are issued.
  const long vox_per_b = (long)a.Xo * a.Yo * a.Zo * (ol_m * ol_m * ol_mz);
  const int olx = ol_ox + pha, oly = ol_oy + phb, oYo = a.Yo * ol_m, oZo = a.Zo * ol_mz;
  const int cob = (nt0 + wn * TN) * 16 + fg * 4;  // this lane's first channel of n-tile j is cob + 16*j
  const bool fast = SIMPLE || (a.vec_ok && !out_planar && (a.Cout & 3) == 0 && (a.mask_c1 & 3) == 0);
  long mrow[TM];   // flat output voxel of row `fr` of m-tile i, or -1
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const unsigned mv = mtab[(wm * TM + i) * 16 + fr];
    const int gx = x0 + (int)(mv & 255), gy = y0 + (int)((mv >> 8) & 255), gz = z0 + (int)((mv >> 16) & 255);
    const bool ok = !(mv >> 24) && gx < a.Xo && gy < a.Yo && gz < a.Zo;
    mrow[i] = ok ? (long)b * vox_per_b + ((long)(gx * ol_m + olx) * oYo + gy * ol_m + oly) * oZo +
                       gz * ol_mz + ol_oz
                 : -1;
  }
  if (part) {  // split reduction: raw sums, [voxel][16*NT_total] fp32 rows; the reduce pass applies the epilogue
    const int cpad = a.NT_total * 16;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (mrow[i] < 0) continue;
      float* prow = part + mrow[i] * cpad + cob;
#pragma unroll
      for (int j = 0; j < TN; ++j)
        if (nt0 + wn * TN + j < a.NT_total)
          *reinterpret_cast<float4*>(prow + 16 * j) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
    return;
  }
  // row base pointers once (64-bit multiply-adds), n-tile offsets are immediates
  E* orow[TM];
  const E* rrow[TM];
  const E* yrow[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long m = mrow[i] < 0 ? 0 : mrow[i];
    orow[i] = reinterpret_cast<E*>(a.out) + m * a.out_ctot + a.out_off + cob;
    rrow[i] = a.res ? reinterpret_cast<const E*>(a.res) + m * a.res_ctot + a.res_off + cob : nullptr;
    yrow[i] = MASK && a.mask_y ? reinterpret_cast<const E*>(a.mask_y) + m * a.mask_ctot + a.mask_off + (cob - a.mask_c0)
                               : nullptr;
  }
  constexpr int FD = TN >= 4 ? 2 : 1;  // operand sets requested ahead of the n-tile being stored
  constexpr int FS = FD + 1;
  float bb[FS][4], ss[FS][4];
  V4 rr[FS][TM], yy[FS][TM];
  auto fetch = [&](int j, int s) __attribute__((always_inline)) {
    const int co0 = cob + 16 * j;
    {
      const int col = (wn * TN + j) * 16 + fg * 4;  // column within the workgroup
      const float4 b4 = *reinterpret_cast<const float4*>(btab + col);
      const float4 s4 = *reinterpret_cast<const float4*>(btab + NTW * 16 + col);
      bb[s][0] = b4.x; bb[s][1] = b4.y; bb[s][2] = b4.z; bb[s][3] = b4.w;
      ss[s][0] = s4.x; ss[s][1] = s4.y; ss[s][2] = s4.z; ss[s][3] = s4.w;
    }
    if (!fast || co0 >= a.Cout) return;
    // LeakyReLU backward of the layer whose output gradient this is (channels [mask_c0, mask_c1)): the
    // multiply by (y > 0 ? 1 : slope) happens in this epilogue, after the accumulation, not in its own pass
    const bool masked = MASK && co0 >= a.mask_c0 && co0 < a.mask_c1;
    if constexpr (EPF) {  // (requested at the top of the last phase: epilogue_prefetch)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        rr[s][i] = pre_rr[j][i];
        yy[s][i] = pre_yy[j][i];
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      rr[s][i] = ct_zero4<T>();
      yy[s][i] = ct_ones4<T>();  // +1: the derivative is 1 outside the mask window
      if (mrow[i] < 0) continue;
      if (a.res && co0 < a.res_c1) rr[s][i] = *reinterpret_cast<const V4*>(rrow[i] + 16 * j);
      if (masked) yy[s][i] = *reinterpret_cast<const V4*>(yrow[i] + 16 * j);
    }
  };
#pragma unroll
  for (int j = 0; j < FD; ++j)
    if (j < TN) fetch(j, j % FS);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int s = j % FS;
    if (j + FD < TN) fetch(j + FD, (j + FD) % FS);
    const int co0 = cob + 16 * j;
    if (co0 >= a.Cout) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (mrow[i] < 0) continue;
      float v[4];
      const bool act_here = a.act && co0 < a.act_c1;  // (act_c1 is a multiple of 4)
      if (fast && a.act == 2) {  // second stage of a split conv: the partial sums in `res` join before the activation
        // (channels >= act_c1 - windows of later convs in a source-grouped stage - stay raw sums: no bias, no activation)
        const float4 rv = ct_cvt4<T>(rr[s][i]);
        const float r4[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float x = acc[i][j][q] + bb[s][q] + a.beta * r4[q];
          if (act_here) x = x > 0.f ? x : x * a.slope;
          v[q] = x * ss[s][q];
        }
        st4<T>(orow[i] + 16 * j, make_float4(v[0], v[1], v[2], v[3]));
        continue;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float x = acc[i][j][q] + bb[s][q];
        if (act_here) x = x > 0.f ? x : x * a.slope;
        v[q] = x * ss[s][q];
      }
      if (fast) {
        float4 o4 = make_float4(v[0], v[1], v[2], v[3]);
        if (a.res) {
          const float4 rv = ct_cvt4<T>(rr[s][i]);
          o4.x += a.beta * rv.x;
          o4.y += a.beta * rv.y;
          o4.z += a.beta * rv.z;
          o4.w += a.beta * rv.w;
        }
        if constexpr (MASK) {  // (bf16: sign test on the raw bits: y > 0 <=> sign clear and not zero)
          const float4 mk = ct_mask4<T>(yy[s][i], a.mask_slope);
          o4.x *= mk.x;
          o4.y *= mk.y;
          o4.z *= mk.z;
          o4.w *= mk.w;
        }
#ifdef WSR_CT_STAMPS
        // (tuning build, ablate & 32: the store pattern of pair-interleaved n-tiles - one 16-byte store per lane for two
        // n-tiles, 64 contiguous bytes per voxel and instruction - with the wrong values: timing only)
        if constexpr (sizeof(E) == 2) {
          if (a.ablate & 32) {
            if (!(j & 1) && j + 1 < TN) {
              float4 d4 = o4;
              d4.x += acc[i][j + 1][0];
              d4.y += acc[i][j + 1][1];
              union { uint4 u; unsigned short h[8]; } pk;
              pk.h[0] = f2bf(o4.x); pk.h[1] = f2bf(o4.y); pk.h[2] = f2bf(o4.z); pk.h[3] = f2bf(o4.w);
              pk.h[4] = f2bf(d4.x); pk.h[5] = f2bf(d4.y); pk.h[6] = f2bf(d4.z); pk.h[7] = f2bf(d4.w);
              *reinterpret_cast<uint4*>(orow[i] + 16 * j + 4 * fg) = pk.u;
              continue;
            }
            if ((j & 1)) continue;
          }
        }
#endif
        if constexpr (TWO) {
          if (a.out2 && co0 >= a.out2_c0) {  // (uniform) this n-tile belongs to the second tensor of the concat's gradient
            st4<T>(reinterpret_cast<E*>(a.out2) + mrow[i] * a.out2_ctot + (co0 - a.out2_c0), o4);
            continue;
          }
        }
        st4<T>(orow[i] + 16 * j, o4);
        continue;
      }
      // general path: planar fp32 output (network boundary), channel tails, unaligned windows
      const long vi = mrow[i] - (long)b * vox_per_b;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (co0 + q >= a.Cout) continue;
        float x = v[q];
        if (out_planar) {
          reinterpret_cast<float*>(a.out)[((long)b * a.Cout + co0 + q) * vox_per_b + vi] = x;
        } else {
          if (a.res && co0 + q < a.res_c1)
            x += a.beta * ldf<T>(reinterpret_cast<const E*>(a.res) + mrow[i] * a.res_ctot + a.res_off + co0 + q);
          if constexpr (MASK) {
            if (co0 + q >= a.mask_c0 && co0 + q < a.mask_c1)
              x *= ldf<T>(reinterpret_cast<const E*>(a.mask_y) + mrow[i] * a.mask_ctot + a.mask_off + (co0 + q - a.mask_c0)) > 0.f
                       ? 1.f : a.mask_slope;
          }
          stf<T>(reinterpret_cast<E*>(a.out) + mrow[i] * a.out_ctot + a.out_off + co0 + q, x);
        }
      }
    }
  }
  CT_STAMP(5);
  CT_STAMP(7);
#ifdef WSR_CT_STAMPS
  if (a.stamps && lane == 0 && (wave == 0 || wave == WAVES - 1)) {  // rows [grid + 2*wg + {0,1}]: first / last wave
    unsigned long long* p = a.stamps + ((size_t)gridDim.x + 2 * blockIdx.x + (wave ? 1 : 0)) * 8;
    p[0] = (unsigned long long)st_dma;
    p[1] = (unsigned long long)st_bar;
    p[2] = (unsigned long long)total_phases;
  }
#endif
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

template <int WM, int WN, int TM, int TN, int TPK, bool MASK = false, class T = BF16, int WK = 1, int SIMPLE = 0>
int launch_ct(CtArgs& a, hipStream_t st) {
  constexpr int WAVES = WM * WN * WK, NTW = WN * TN;
  const int taps = a.KX * a.KY * a.KZ;
  constexpr int M = WM * TM * 16;  // table sizes follow the MFMA rows; the tile volume may be smaller
  if (a.TX * a.TY * a.TZ > M) return WSR_EUNSUPPORTED;
  if (a.in2 || a.out2) {  // two-tensor concat (wsr_epilogue_t.in2 / wsr_dgrad_opts_t.dx2): the 9-n-tile instantiations only
    constexpr int CKH = T::EPP * (4 / TPK);
    if (!ct_two_src(WN, TN) || a.ups || a.nphase == 4) return WSR_EUNSUPPORTED;
    if (a.in2 && (a.in2_ctot % T::EPP || a.in2_chunk < 1 || a.in2_chunk * CKH >= a.cin_valid)) return WSR_EINVAL;
    if (a.out2 && (a.out_planar || !a.vec_ok || (a.Cout & 3) || a.res || a.mask_y || (a.out2_c0 & 15) || (a.out2_ctot & 3) ||
                   a.out2_c0 <= 0 || a.out2_c0 >= a.Cout || a.Cout - a.out2_c0 > a.out2_ctot))
      return WSR_EUNSUPPORTED;
    a.ws = nullptr;  // (never split: the split-reduction pass knows one produced tensor)
    a.ws_bytes = 0;
  }
  if (a.sx < 1) a.sx = a.sy = a.sz = 1;
  if (a.il_m < 1) a.il_m = 1;
  if (a.ol_m < 1) a.ol_m = 1;
  if (a.ol_mz < 1) a.ol_mz = 1;
  if (a.nphase != 4) a.nphase = 1;
  if ((a.il_m > 1 || a.ol_m > 1 || a.ol_mz > 1 || a.nphase > 1) && (a.ups || a.sx != 1 || a.sy != 1 || a.sz != 1)) return WSR_EUNSUPPORTED;
  if constexpr (SIMPLE) {  // what the kernel's SIMPLE form takes for granted (the caller goes on to the general instantiation)
    if (a.sx != 1 || a.sy != 1 || a.sz != 1 || a.il_m != 1 || a.ol_m != 1 || a.ol_mz != 1 || a.nphase != 1 || a.ups ||
        a.il_ox || a.il_oy || a.ol_ox || a.ol_oy || a.ol_oz || a.out_planar || !a.vec_ok || (a.Cout & 3) || (a.mask_c1 & 3))
      return WSR_EUNSUPPORTED;
    if (SIMPLE == 1 && (a.chan_scale || a.in2 || a.out2)) return WSR_EUNSUPPORTED;
  }
  const int L = ((a.TX - 1) * a.sx + a.KX) * ((a.TY - 1) * a.sy + a.KY) * ((a.TZ - 1) * a.sz + a.KZ);
  if (L > 65535) return WSR_EUNSUPPORTED;
  a.nts = (taps + TPK - 1) / TPK;
  a.NT_total = (a.Cout + 15) / 16;
  a.ngroups = (a.NT_total + NTW - 1) / NTW;
  constexpr int PL = 4 / TPK;
  constexpr bool VM = TPK == 2;
  a.P = VM ? round_up(L * 32, 1024) : round_up(L * 16, 1024);  // whole 1 KB DMA units; == 0 (mod 256)
  a.off_mtab = 0;
  a.off_htab = M * 4;
  a.off_ttab = round_up(a.off_htab + M * 2, 16);
  a.off_btab = round_up(a.off_ttab + a.nts * TPK * 4, 16);
  a.off_xs = round_up(a.off_btab + 2 * NTW * 16 * 4, 1024);
  // two activation buffers when that still leaves room for weight stages of >= 2 K-steps
  int ts_max = 0;
  // (one chunk: the second activation buffer would never be filled - the LDS it frees lets a second workgroup share
  // the CU, whose prologue and epilogue then overlap this one's main loop: terrain convs, 3-channel inputs)
  const int xb_first = WSR_ENV_INT("WSR_CT_XBUFS", (a.nchunks == 1 || (NTW == 1 && WSR_ENV_SET("WSR_CT_N16_ONEBUF"))) ? 1 : 2);  // env: tuning aid
  // Two workgroups per CU for the 32-wide launches (WSR_CT_DIET=1, round 6): their 125 registers leave room for four
  // waves per SIMD, so a workgroup that keeps to half the LDS (one activation buffer, shorter weight stages) shares
  // its CU with a second one whose prologue / epilogue / DMA waits then run under this one's K-steps.
  // Measured (profiles/r06_b_ab_diet.txt, single launches, same device): at batch 4 (1 024 tiles = four rounds) the
  // 32 -> 32 / 64 -> 32 / 96 -> 32 growth stages take 55.8 / 78.2 / 102.6 us as one 128 KB workgroup per CU and
  // 45.1 / 69.8 / 98.8 us as two 67 KB ones; at batch 1 (256 tiles: nothing to share a CU with) the single buffer costs
  // 14.6 -> 15.6, 20.5 -> 23.0, 28.3 -> 33.9 us, and 512 tiles of 256 voxels (WSR_CT_NARROW_M=256) are no better than
  // that (16.0 / 24.9 / 34.3): the diet is taken from two rounds of workgroups on (WSR_CT_DIET=0 / 1 forces it).
  const long nwg_all = (long)a.B * ((a.Xo + a.TX - 1) / a.TX) * ((a.Yo + a.TY - 1) / a.TY) * ((a.Zo + a.TZ - 1) / a.TZ) *
                       ((a.Cout + 16 * NTW - 1) / (16 * NTW)) * (a.nphase == 4 ? 4 : 1);
  // (only where it was measured - the trunk's SIMPLE 32-wide instantiations - and only when half the LDS still holds one
  // activation buffer and weight stages of >= 3 K-steps: a strided 32-wide conv of the discriminator, whose 105 KB halo image
  // does not fit, would otherwise have been turned away to smaller tiles)
  bool diet = SIMPLE == 1 && NTW <= 2 && WAVES == 8 && a.nchunks > 1 && WSR_ENV_INT("WSR_CT_DIET", nwg_all >= 512 ? 1 : 0);
  if (diet) {
    const int room = 80 * 1024 - (a.off_xs + (VM ? 1 : PL) * a.P);
    const int ts_diet = room / (2 * NTW * 1024);
    if (ts_diet < 3 && ts_diet < a.nts) diet = false;
  }
  const int lds_cap = diet ? 80 * 1024 : 160 * 1024;
  for (a.xbufs = xb_first; a.xbufs >= 1; --a.xbufs) {
    a.off_ws = a.off_xs + a.xbufs * (VM ? 1 : PL) * a.P;
    const int avail = lds_cap - a.off_ws;
    ts_max = avail / (2 * NTW * 1024);
    const int cap_kb = WSR_ENV_INT("WSR_WSTAGE_KB", 48);  // tuning aid
    const int cap = cap_kb / NTW > 0 ? cap_kb / NTW : 1;  // <= 48 KB per weight stage (measured: up-convs +7 %, others flat)
    if (ts_max > cap) ts_max = cap;
    // (weight stages shorter than 3 K-steps cost more in barriers than the activation prefetch saves)
    if (ts_max >= 3 || ts_max >= a.nts || a.xbufs == 1) break;
  }
  if (ts_max < 1) return WSR_EUNSUPPORTED;
  if ((VM ? (L + 31) / 32 : ((L + 63) / 64) * PL) > ct_xk(WAVES, TM) * WAVES) return WSR_EUNSUPPORTED;  // XK units per wave
  {  // 32-bit element offsets relative to the halo's first x-plane: (stored planes the halo spans + 1) x plane size
    const long lx = ((long)(a.TX - 1) * a.sx + a.KX + 1) * a.il_m + 1;
    if (lx * a.Yi * a.il_m * a.Zi * a.in_ctot >= 0xFFFFFFFFL) return WSR_EUNSUPPORTED;
  }
  const int nph = (a.nts + ts_max - 1) / ts_max;
  a.TS = (a.nts + nph - 1) / nph;  // balanced stages
  size_t lds = (size_t)a.off_ws + (size_t)2 * a.TS * NTW * 1024;
  if (WK > 1) {  // the K-step shares' partial sums meet in the (then free) activation / weight buffers
    const size_t red = (size_t)a.off_xs + (size_t)(WK - 1) * WM * WN * TM * TN * 1024;
    if (red > lds) lds = red;
  }
  if (lds > 160 * 1024) return WSR_EUNSUPPORTED;
  a.xs_stage = -1;
  a.xs_units = 0;
  if (TN == 9 && a.xbufs == 1 && a.nchunks > 1 && VM && a.KX >= 2 && a.sx == 1 && !WSR_ENV_SET("WSR_CT_NOXSPLIT")) {
    const int Ly_ = (a.TY - 1) * a.sy + a.KY, Lz_ = (a.TZ - 1) * a.sz + a.KZ;
    const int plane_vox = Ly_ * Lz_;                       // halo voxels of one x-plane
    const int early_vox = (a.KX - 1) * plane_vox;          // planes [0, KX-1)
    const int ks_last = ((a.KX - 1) * a.KY * a.KZ + TPK - 1) / TPK;  // first K-step whose taps are all in column KX-1
    const int nst = (a.nts + a.TS - 1) / a.TS;
    const int se = (ks_last + a.TS - 1) / a.TS;            // first weight stage made of such K-steps only
    const int ks_kx1 = (a.KY * a.KZ) / TPK;                // first K-step that touches tap column 1
    // conditions: the early part is whole DMA units; a stage boundary exists inside the last tap column; the first
    // stage of a chunk stays inside tap column 0 (it runs before the late part is published) and reads only planes
    // < TX <= KX-1 ... the late part holds planes >= KX-1, read from tap column max(1, KX-TX) on
    if (early_vox % 32 == 0 && se < nst && a.TS <= ks_kx1 && a.TX <= a.KX - 1) {
      a.xs_stage = se;
      a.xs_units = early_vox / 32;
    }
  }
  a.tiles_x = (a.Xo + a.TX - 1) / a.TX;
  a.tiles_y = (a.Yo + a.TY - 1) / a.TY;
  a.tiles_z = (a.Zo + a.TZ - 1) / a.TZ;
  a.ntiles = a.B * a.tiles_x * a.tiles_y * a.tiles_z;
  {  // fdiv operand range: n * d < 2^32 for the workgroup-index decode (the table divisions have n < 65536)
    int dmax = a.ngroups;
    if (a.tiles_z > dmax) dmax = a.tiles_z;
    if (a.tiles_y > dmax) dmax = a.tiles_y;
    if (a.tiles_x > dmax) dmax = a.tiles_x;
    if ((long)a.ntiles * a.nphase * a.ngroups * dmax >= (1l << 32)) return WSR_EUNSUPPORTED;
  }
  a.mg_TZ = fdiv_magic(a.TZ); a.mg_TY = fdiv_magic(a.TY);
  a.mg_Lz = fdiv_magic((a.TZ - 1) * a.sz + a.KZ); a.mg_Ly = fdiv_magic((a.TY - 1) * a.sy + a.KY);
  a.mg_KZ = fdiv_magic(a.KZ); a.mg_KY = fdiv_magic(a.KY);
  a.mg_ng = fdiv_magic(a.ngroups); a.mg_tz = fdiv_magic(a.tiles_z);
  a.mg_ty = fdiv_magic(a.tiles_y); a.mg_tx = fdiv_magic(a.tiles_x);
  // register budget: (TM+TN)*8 fragment + TM*TN*4 accumulator VGPRs; four-wave workgroups have 512 per wave
  constexpr bool PIPE = TN <= 7 || WAVES <= 4;
  if (a.mask_y && !MASK) return WSR_EUNSUPPORTED;
  if (std::is_same<T, F32>::value) a.ws = nullptr;  // (the split-reduction second pass writes bf16)
  auto kern = conv_tile_kernel<WM, WN, TM, TN, TPK, PIPE, MASK, T, WK, SIMPLE>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
#ifdef WSR_CT_STAMPS
  a.ablate = WSR_ENV_INT("WSR_CT_ABL", 0);
  a.stamps = getenv("WSR_CT_STAMPS_PTR") ? (unsigned long long*)strtoull(getenv("WSR_CT_STAMPS_PTR"), nullptr, 0) : nullptr;
#endif
  // static priority for the younger half of the workgroup: measured again on the final kernels - +1.2 % on the
  // 144-wide 5x5x5 tile, but -1.5 ... -4 % on every other instantiation (128-wide: 109.7 -> 106.6 us, narrow:
  // 24.0 -> 23.5 us, 160..224-wide: 60.0 -> 58.2 us) since the operand requests moved ahead of the MFMAs
  a.prio = WSR_ENV_INT("WSR_CT_PRIO", TN == 9 ? 1 : 0);
  // Few workgroups and a long reduction (the deep layers of the discriminator: 16..128 workgroups walking
  // 16..32 chunks x 27..48 taps one after the other, 50-90 us at a few per cent of the chip): split the chunks over
  // ksplit times as many workgroups; the partial sums go through the caller's workspace.
  a.ksplit = 1;
  a.part = nullptr;
  const int wg = a.ntiles * a.nphase * a.ngroups;
  if (SIMPLE && a.ws && wg <= 128 && a.nchunks >= 4) return WSR_EUNSUPPORTED;  // (might split the reduction: general form)
  if (a.ws && wg <= 128 && a.nchunks >= 4 && a.nphase == 1 && a.ol_m == 1 && a.ol_mz == 1 && !a.res && !a.mask_y && !a.chan_scale &&
      !a.out_planar && a.act <= 1 && a.act_c1 == 0x7FFFFFFF && (a.Cout & 3) == 0 && a.vec_ok && !WSR_ENV_SET("WSR_CT_NOSPLITK")) {
    int ks = 256 / wg;
    if (ks > a.nchunks / 2) ks = a.nchunks / 2;
    if (ks >= 2) {
      const int cps = (a.nchunks + ks - 1) / ks;
      ks = (a.nchunks + cps - 1) / cps;
      const long stride = (long)a.B * a.Xo * a.Yo * a.Zo * a.NT_total * 16;
      if (ks >= 2 && (long)ks * stride * 4 <= a.ws_bytes) {
        a.ksplit = ks; a.cps = cps; a.part = (float*)a.ws; a.part_stride = stride;
      }
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(wg * a.ksplit)), dim3(WAVES * 64), lds, st, a);
  WSR_LAUNCH_CHECK();
  if (a.ksplit > 1) return wsr_ct_splitk_reduce(a, st);
  return 0;
}

// choose the spatial tile: at most M voxels, z (the contiguous axis) kept whole when it is short
static void pick_tile(CtArgs& a, int M) {
  int tz;
  if (a.Zo <= 16) tz = a.Zo;
  else if (a.Zo % 16 == 0 || a.Zo > 64) tz = 16;
  else tz = 8;
  // no taps along z (the z-folded last conv, 5x5x1): a flat tile has no z halo and a smaller x-y one
  // (8x16x4: 1.9x its 512 voxels instead of 3x for 4x8x16) - these launches are bound by the halo re-reads
  if (a.KZ == 1 && a.KX * a.KY > 1 && tz > 4 && !WSR_ENV_SET("WSR_CT_NOFLAT")) tz = 4;
  while (tz > M) tz >>= 1;
  const int rest = M / tz;
  int tx = 1, ty = 1;
  if ((rest & (rest - 1)) == 0) {
    while (tx * ty < rest) {  // power of two: balanced split, y first
      if (ty <= tx) ty <<= 1; else tx <<= 1;
    }
  } else if (WSR_ENV_SET("WSR_CT_SQUARE_TILES")) {  // (rounds 1-4: the most nearly square x-y tile, whatever the extents)
    while ((tx + 1) * (tx + 1) <= rest) ++tx;
    ty = rest / tx;
  } else {
    // z extents such as the reference's 10 levels leave an x-y budget that is no power of two (512 / 10 = 51): take the
    // split that covers THIS volume with the fewest tiles - 16 x 16 x 10 (the LR patches of the shipped configurations,
    // config/wind_field_GAN_3D_config_cluster.ini:42-47) needs 3 x 3 = 9 tiles of 7 x 7 x 10 but only 2 x 3 = 6 of 8 x 6 x 10:
    // a third of the trunk's workgroups were padding - and among equals the one with the smallest halo image.
    // Candidates stay inside what the tables can express and what the square picker's launches could afford: tile
    // coordinates are packed in 8 bits each (mtab: ox | oy << 8 | oz << 16), so cx, cy <= 255 and no larger than the
    // extent itself (a needle through a long thin volume would wrap them), and a halo image more than 1.3 x the most
    // nearly square tile's is refused (it would fail the LDS check and fall back to the generic kernel).
    const int sxs = a.sx > 0 ? a.sx : 1, sys_ = a.sy > 0 ? a.sy : 1;
    int sq = 1;
    while ((sq + 1) * (sq + 1) <= rest) ++sq;
    const long halo_sq = (long)((sq - 1) * sxs + a.KX) * ((rest / sq - 1) * sys_ + a.KY);
    const int cx_max = a.Xo < 255 ? (a.Xo > 0 ? a.Xo : 1) : 255, cy_max = a.Yo < 255 ? (a.Yo > 0 ? a.Yo : 1) : 255;
    long best_tiles = -1, best_halo = 0;
    for (int cx = 1; cx <= rest && cx <= cx_max; ++cx) {
      int cy = rest / cx;
      if (cy < 1) break;
      if (cy > cy_max) cy = cy_max;
      const long tiles = (long)((a.Xo + cx - 1) / cx) * ((a.Yo + cy - 1) / cy);
      const long halo = (long)((cx - 1) * sxs + a.KX) * ((cy - 1) * sys_ + a.KY);
      if (halo * 10 > halo_sq * 13) continue;
      if (best_tiles < 0 || tiles < best_tiles || (tiles == best_tiles && halo < best_halo)) {
        best_tiles = tiles; best_halo = halo; tx = cx; ty = cy;
      }
    }
    if (best_tiles < 0) { tx = sq < cx_max ? sq : cx_max; ty = rest / sq < cy_max ? rest / sq : cy_max; }
  }
  a.TX = tx; a.TY = ty; a.TZ = tz;
}


}  // namespace
