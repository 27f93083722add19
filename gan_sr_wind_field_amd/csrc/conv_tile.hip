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
#include <cstdlib>

#include "conv_tile_impl.h"

int wsr_ct_run_n144(CtArgs& a, int tpk, hipStream_t st);    // conv_tile_n144.hip
int wsr_ct_run_n128(CtArgs& a, int tpk, hipStream_t st);    // conv_tile_n128.hip
int wsr_ct_run_narrow(CtArgs& a, int tpk, hipStream_t st);  // conv_tile_narrow.hip
int wsr_ct_run_wide(CtArgs& a, int tpk, hipStream_t st);    // conv_tile_wide.hip
int wsr_ct_run_masked(CtArgs& a, int tpk, hipStream_t st);  // conv_tile_masked.hip
int wsr_ct_run_narrow_masked(CtArgs& a, int tpk, hipStream_t st);  // conv_tile_narrow_masked.hip
int wsr_ct_run_small(CtArgs& a, int tpk, hipStream_t st);          // conv_tile_small.hip
int wsr_ct_run_tm3(CtArgs& a, int tpk, hipStream_t st);            // conv_tile_tm3.hip
int wsr_ct_run_narrow_wk(CtArgs& a, int tpk, hipStream_t st);      // conv_tile_narrow_wk.hip
int wsr_ct_run_simple_narrow(CtArgs& a, int tpk, int tm3, hipStream_t st);  // conv_tile_simple_narrow.hip
int wsr_ct_run_simple_n128(CtArgs& a, int tpk, int tm3, hipStream_t st);    // conv_tile_simple_n128.hip
int wsr_ct_run_simple_mid(CtArgs& a, int tpk, hipStream_t st);              // conv_tile_simple_mid.hip
int wsr_ct_run_simple_small(CtArgs& a, int tpk, hipStream_t st);            // conv_tile_simple_small.hip
long wsr_ct_tiles(const CtArgs& a, int rows);                      // conv_tile_tm3.hip
int wsr_ct_run_strided(CtArgs& a, int tpk, hipStream_t st);        // conv_tile_strided.hip
int wsr_ct_run_f32(CtArgs& a, int tpk, hipStream_t st);            // conv_tile_f32*.hip
#ifdef WSR_TUNING
int wsr_ct_run_w4(CtArgs& a, int tpk, int which, hipStream_t st);  // conv_tile_w4.hip (make TUNING=1)
#endif

namespace {

int dispatch_ct(CtArgs& a, int tpk, hipStream_t st) {
  const int N = a.Cout;
  if (a.f32) return wsr_ct_run_f32(a, tpk, st);  // (stride 1 only: the entry points checked)
  if ((a.sx | a.sy | a.sz) != 1) return wsr_ct_run_strided(a, tpk, st);
#ifdef WSR_TUNING
  if (WSR_ENV_SET("WSR_CT_W4")) {  // tuning switch: four-wave workgroups
    if ((long)a.B * a.Xo * a.Yo * a.Zo >= 128L * 512) {
      const int rc = wsr_ct_run_w4(a, tpk, WSR_ENV_RAW("WSR_CT_W4"), st);
      if (rc != WSR_EUNSUPPORTED) return rc;
    }
  }
#endif
  // small volumes (< 128 tiles of 512 voxels): 128-voxel tiles where an instantiation exists
  // (SIMPLE: instantiations with the general forms' run-time switches folded away, for the plain stride-1 launches of the
  // trunk - conv_tile_impl.h; each returns WSR_EUNSUPPORTED for anything else)
  const int simple = WSR_ENV_INT("WSR_CT_SIMPLE", 1);
  if ((long)a.B * a.Xo * a.Yo * a.Zo < 128L * 512 && !WSR_ENV_SET("WSR_CT_NOSMALL")) {
    if (simple & 1) {
      const int rc = wsr_ct_run_simple_small(a, tpk, st);
      if (rc != WSR_EUNSUPPORTED) return rc;
    }
    const int rc = wsr_ct_run_small(a, tpk, st);
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  // 384-voxel tiles where they need fewer (rounds of 256 workgroups) x (MFMA rows per workgroup) than 512-voxel ones:
  // single-round launches that leave CUs idle (conv_tile_tm3.hip: the cluster configuration's trunk, 192 -> 256
  // workgroups of three quarters the length)
  int tm3 = 0;
  if (tpk == 2 && ((N > 16 && N <= 32) || (N > 64 && N <= 128 && !a.mask_y)) && a.nphase != 4 && !a.ups && !WSR_ENV_SET("WSR_CT_NO_TM3")) {
    const long n4 = wsr_ct_tiles(a, 512), n3 = wsr_ct_tiles(a, 384);
    tm3 = ((n3 + 255) / 256) * 3 < ((n4 + 255) / 256) * 4;
  }
  // the source-grouped stages of a dense block's forward (act = 2 with a partial activation window; engine.conv_dense)
  if ((simple & 1) && tpk == 2 && N > 32 && N <= 96 && !a.mask_y && a.act == 2 && a.act_c1 != 0x7FFFFFFF && a.nphase != 4 &&
      !a.ups && !WSR_ENV_SET("WSR_CT_NO_MID")) {
    const int rc = wsr_ct_run_simple_mid(a, tpk, st);
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  if ((simple & 1) && tpk == 2 && N > 16 && N <= 32 && !(WSR_ENV_INT("WSR_CT_NARROW_WK", 0))) {
    const int rc = wsr_ct_run_simple_narrow(a, tpk, tm3, st);
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  if ((simple & 1) && tpk == 2 && N > 64 && N <= 128 && !a.mask_y && !WSR_ENV_SET("WSR_CT_NO_N128")) {
    const int rc = wsr_ct_run_simple_n128(a, tpk, tm3, st);
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  if (tm3) {
    const int rc = wsr_ct_run_tm3(a, tpk, st);
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  if (tpk == 2 && N > 16 && N <= 32 && a.nphase != 4 && WSR_ENV_INT("WSR_CT_NARROW_WK", 0)) {  // (tuning: 16-wave K-step shares)
    const int rc = wsr_ct_run_narrow_wk(a, tpk, st);
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  if (a.mask_y) return N <= 64 ? wsr_ct_run_narrow_masked(a, tpk, st) : wsr_ct_run_masked(a, tpk, st);
  if (N <= 64) return wsr_ct_run_narrow(a, tpk, st);
  // (short reductions - the z-folded last conv's input gradient, 16 channels x 25 taps - stay on the generic
  // wide tiles: the 512-voxel N = 144 instantiation is kept for the 5x5x5 144 -> 144 conv it is tuned and
  // profiled for)
  // (a SIMPLE == 2 instantiation of the 144-wide tile - channel scale and two-tensor concat kept - measured 7.13 -> 7.15-7.17 ms
  // on the 5x5x5 conv: its launches are 7 ms of main loop, and that loop did not get shorter; not shipped)
  if (N == 144 && (long)a.nchunks * a.KX * a.KY * a.KZ >= 256) return wsr_ct_run_n144(a, tpk, st);
  if (N > 64 && N <= 128 && !WSR_ENV_SET("WSR_CT_NO_N128")) {  // (the env switch is a tuning aid)
    const int rc = wsr_ct_run_n128(a, tpk, st);
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  if (N <= 256) return wsr_ct_run_wide(a, tpk, st);
  return WSR_EUNSUPPORTED;
}

// ---- fragment-order filter packing ----------------------------------------------------------
// out[chunk][kstep][ntile][lane][e]: lane (i = lane&15, g = lane>>4), octet g -> tap kstep*TPK + g/PL,
// channel chunk*CK + (g%PL)*8 + e.  transpose = 0: rows n = Cout, reduction c = Cin (forward);
// transpose = 1: rows n = Cin, reduction c = Cout, taps flipped (input gradient).
__device__ uint4 g_zero16 = {0u, 0u, 0u, 0u};

// K-step shape of a reduction over `redp` stored channels (a multiple of the piece size `epp`): TPK taps x 32/TPK
// bytes... in channels: bf16 (epp 8) 32 / 16 / 8 per tap, fp32 (epp 4) 16 / 8 / 4
static inline int ct_tpk(int redp, int taps, int epp) {
  if (taps == 1) return redp % (4 * epp) == 0 ? 1 : (redp % (2 * epp) == 0 ? 2 : 4);
  return redp % (2 * epp) == 0 ? 2 : 4;
}

template <class T>
__global__ void pack_frag_kernel(const float* __restrict__ w, typename T::elem* __restrict__ out, int Cout, int Cin,
                                 int KX, int KY, int KZ, int transpose, int TPK, int nchunks, int nts, int NT_total) {
  constexpr int EPP = T::EPP;
  const int PL = 4 / TPK, CK = EPP * PL;
  const int taps = KX * KY * KZ;
  const long total = (long)nchunks * nts * NT_total * 64 * EPP;
  const int rows = transpose ? Cin : Cout, red = transpose ? Cout : Cin;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int e = (int)(idx % EPP);
    const int lane = (int)((idx / EPP) & 63);
    long q = idx / (64 * EPP);
    const int nt = (int)(q % NT_total); q /= NT_total;
    const int ts = (int)(q % nts);
    const int chunk = (int)(q / nts);
    const int i = lane & 15, g = lane >> 4;
    const int tap = ts * TPK + g / PL;
    const int c = chunk * CK + (g % PL) * EPP + e;
    const int n = nt * 16 + i;
    float v = 0.f;
    if (tap < taps && c < red && n < rows) {
      if (!transpose) {
        v = w[((long)n * Cin + c) * taps + tap];
      } else {
        v = w[((long)c * Cin + n) * taps + (taps - 1 - tap)];
      }
    }
    stf<T>(out + idx, v);
  }
}

// The job-table form for any element type, one element per thread (gather): what the fp32 programs use - their
// packing is < 0.1 % of a step; the bf16 programs take the LDS-staged kernel below.
template <class T>
__global__ void pack_frag_multi_gather_kernel(const wsr_pack_job_t* __restrict__ jobs) {
  constexpr int EPP = T::EPP;
  const wsr_pack_job_t j = jobs[blockIdx.y];
  const int taps = j.KX * j.KY * j.KZ;
  const bool part = j.red_total > 0;  // one source of a stacked dense-block filter
  const int rows = part ? (j.transpose ? j.c_n : j.rows_total) : (j.transpose ? j.Cin : j.Cout);
  const int red = part ? j.red_total : (j.transpose ? j.Cout : j.Cin);
  const int redp = (red + EPP - 1) / EPP * EPP;
  const int TPK = taps == 1 ? (redp % (4 * EPP) == 0 ? 1 : (redp % (2 * EPP) == 0 ? 2 : 4)) : (redp % (2 * EPP) == 0 ? 2 : 4);
  const int PL = 4 / TPK, CK = EPP * PL;
  const int nts = (taps + TPK - 1) / TPK, NT_total = (rows + 15) / 16;
  const int chunk0 = part && j.transpose ? j.red_off / CK : 0;
  const int nchunks = part && j.transpose ? (j.Cout + CK - 1) / CK : (redp + CK - 1) / CK;
  const int nt0 = part && !j.transpose ? j.row_off / 16 : 0;
  const int ntl = part && !j.transpose ? (j.Cout + 15) / 16 : NT_total;
  const int src_rows = j.transpose ? (part ? j.c_n : j.Cin) : j.Cout;
  const int src_red = j.transpose ? j.Cout : (part ? j.c_n : j.Cin);
  const int c_lo = part ? j.c_lo : 0;
  typename T::elem* __restrict__ out = reinterpret_cast<typename T::elem*>(j.out);
  const long total = (long)nchunks * nts * ntl * 64 * EPP;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int e = (int)(idx % EPP);
    const int lane = (int)((idx / EPP) & 63);
    long q = idx / (64 * EPP);
    const int nt = (int)(q % ntl); q /= ntl;
    const int ts = (int)(q % nts);
    const int chunk = (int)(q / nts);
    const int i = lane & 15, g = lane >> 4;
    const int tap = ts * TPK + g / PL;
    const int c = chunk * CK + (g % PL) * EPP + e;
    const int n = nt * 16 + i;
    float v = 0.f;
    if (tap < taps && c < src_red && n < src_rows) {
      const long tap_src = j.transpose ? taps - 1 - tap : tap;
      v = j.transpose ? j.w[((long)c * j.Cin + c_lo + n) * taps + tap_src] : j.w[((long)n * j.Cin + c_lo + c) * taps + tap_src];
    }
    stf<T>(out + ((((long)(chunk0 + chunk) * nts + ts) * NT_total + nt0 + nt) * 64 + lane) * EPP + e, v);
  }
}

// One block row (blockIdx.y) per job.  An item = one (chunk, n-tile, range of <= 32 taps) of the job's sub-block: its
// 16 rows x CK reduction channels x taps are contiguous runs of the master filter (CK * taps floats per row forward,
// 16 * taps per reduction channel transposed); they are read coalesced into LDS and leave as whole 16-byte fragments.
// (The first form gathered one float per 2-byte store, `taps` floats apart: 4.5x the filter bytes fetched.  Measured
// and dropped: one flat grid over all jobs' items with the prefix sums of the item counts built per workgroup - the
// prefix costs more than the ~60 000 mostly empty workgroups of a network's table - and the next item's loads
// issued under the fragment writes - 228 registers, spills.)
constexpr int PK_TRMAX = 32;

struct PkGeom {
  int taps, TPK, pls, cks, CK, nts, NT_total, chunk0, nchunks, nt0, ntl, src_rows, src_red, c_lo, TR, ntr, items;
};
__device__ __forceinline__ PkGeom pk_geom(const wsr_pack_job_t j) {
  PkGeom g;
  g.taps = j.KX * j.KY * j.KZ;
  const bool part = j.red_total > 0;  // one source of a stacked dense-block filter
  const int rows = part ? (j.transpose ? j.c_n : j.rows_total) : (j.transpose ? j.Cin : j.Cout);
  const int red = part ? j.red_total : (j.transpose ? j.Cout : j.Cin);
  const int redp = (red + 7) / 8 * 8;
  g.TPK = g.taps == 1 ? (redp % 32 == 0 ? 1 : (redp % 16 == 0 ? 2 : 4)) : (redp % 16 == 0 ? 2 : 4);
  g.pls = g.TPK == 1 ? 2 : (g.TPK == 2 ? 1 : 0);  // PL = 4 / TPK = 1 << pls channel octets per tap, CK = 8 << pls
  g.cks = 3 + g.pls;
  g.CK = 1 << g.cks;
  const int tps = 2 - g.pls;                      // TPK = 1 << tps
  g.nts = (g.taps + g.TPK - 1) >> tps;
  g.NT_total = (rows + 15) / 16;
  // the sub-block this job writes: chunks [chunk0, chunk0 + nchunks), n-tiles [nt0, nt0 + ntl)
  g.chunk0 = part && j.transpose ? j.red_off >> g.cks : 0;
  g.nchunks = part && j.transpose ? j.Cout >> g.cks : (redp + g.CK - 1) >> g.cks;
  g.nt0 = part && !j.transpose ? j.row_off / 16 : 0;
  g.ntl = part && !j.transpose ? (j.Cout + 15) / 16 : g.NT_total;
  g.src_rows = j.transpose ? (part ? j.c_n : j.Cin) : j.Cout;  // rows the source block has
  g.src_red = j.transpose ? j.Cout : (part ? j.c_n : j.Cin);   // reduction channels it has
  g.c_lo = part ? j.c_lo : 0;
  // tap ranges: TR padded taps (a multiple of TPK) at a time
  const int ptaps = g.nts << tps;
  g.TR = ptaps < PK_TRMAX ? ptaps : PK_TRMAX;
  g.ntr = (ptaps + g.TR - 1) / g.TR;
  g.items = g.nchunks * g.ntl * g.ntr;
  return g;
}

__global__ __launch_bounds__(256) void pack_frag_multi_kernel(const wsr_pack_job_t* __restrict__ jobs, int vec_on) {
  __shared__ float sh[512 * 17 > 256 * (PK_TRMAX + 1) ? 512 * 17 : 256 * (PK_TRMAX + 1)];
  const wsr_pack_job_t jrec = jobs[blockIdx.y];
  const PkGeom g = pk_geom(jrec);
  struct { const float* w; void* out; int transpose, Cin; } j = {jrec.w, jrec.out, jrec.transpose, jrec.Cin};  // (scalars)
  const int taps = g.taps, TR = g.TR, pitch = TR + 1, cks = g.cks, CK = g.CK, pls = g.pls, tps = 2 - g.pls;
  const int nrow = 16 << cks;  // LDS rows: forward i * CK + cl, transposed cl * 16 + i  (source order)
  const float* __restrict__ w = j.w;
  uint4* __restrict__ out = reinterpret_cast<uint4*>(j.out);
  const int t = threadIdx.x;
  // (index arithmetic by shifts only: with runtime divisions the kernel was bound by them, not by memory)
  // loads: thread (tx, ty) reads tap t0 + tx of rows ty, ty + RP, ...; one column when a range is one tap (1x1x1)
  const int tcs = TR == 1 ? 0 : 5;
  const int tx = t & ((1 << tcs) - 1), ty = t >> tcs, RP = 256 >> tcs;
  for (int item = blockIdx.x; item < g.items; item += gridDim.x) {
    const int tr = item % g.ntr;
    const int q = item / g.ntr;
    const int nt = q % g.ntl, chunk = q / g.ntl;
    const int t0 = tr * TR;  // first (destination-order) tap of the range
    const int tap = t0 + tx;
    const bool tap_ok = tx < TR && tap < taps;
    const long tap_src = j.transpose ? taps - 1 - tap : tap;
    __syncthreads();
    // Round 4, 16-byte loads: the LDS rows in source order are O runs of R rows x taps floats, each run CONTIGUOUS in the
    // master filter (forward: the CK reduction channels of one output channel; transposed: the 16 rows of one reduction
    // channel) - read as float4 by 16 threads per run and kept in LDS as they are, [row][tap] without padding and without
    // the tap flip (one 16-byte LDS write per load; scattered into the padded, flipped image of the one-float form the
    // 4-byte writes' bank conflicts cost more than the loads saved); the fragment pass below indexes accordingly.  The
    // one-float form moved 0.7 TB/s.  Needs all taps in one range and 16-byte aligned runs whose valid part is whole
    // float4s (else: the one-float form, per item).
    const int R = j.transpose ? 16 : CK, O = nrow / R;
    const int inner0 = j.transpose ? nt * 16 : chunk * CK, inner_lim = j.transpose ? g.src_rows : g.src_red;
    const int outer0 = j.transpose ? chunk * CK : nt * 16, outer_lim = j.transpose ? g.src_red : g.src_rows;
    int vin = inner_lim - inner0;
    vin = vin < 0 ? 0 : (vin > R ? R : vin);  // valid rows of a run
    const bool vec = vec_on && g.ntr == 1 && taps > 1 && ((vin * taps) & 3) == 0 && (((long)j.Cin * taps) & 3) == 0 &&
                     ((g.c_lo * taps) & 3) == 0 && (((size_t)w) & 15) == 0;
    if (vec) {
      const int q4 = (vin * taps) >> 2, rt4 = (R * taps) >> 2;  // float4s of a run: valid, whole (R is 8, 16 or 32)
      for (int o = t >> 4; o < O; o += 16) {
        const bool ook = outer0 + o < outer_lim;
        const float* run = w + ((long)(outer0 + o) * j.Cin + g.c_lo + inner0) * taps;
        for (int f0 = t & 15; f0 < rt4; f0 += 16 * 8) {
          float4 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int f4 = f0 + 16 * u;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ook && f4 < q4) v[u] = *reinterpret_cast<const float4*>(run + 4 * f4);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int f4 = f0 + 16 * u;
            if (f4 < rt4) *reinterpret_cast<float4*>(sh + (o * rt4 + f4) * 4) = v[u];
          }
        }
      }
    } else
    // (sixteen loads in flight per thread: left as a plain loop the compiler waits for every float before the next)
    for (int r0 = ty; r0 < nrow; r0 += RP * 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int r = r0 + u * RP;
        const int i = j.transpose ? (r & 15) : r >> cks, cl = j.transpose ? (r >> 4) : r & (CK - 1);
        const int n = nt * 16 + i, c = chunk * CK + cl;
        v[u] = 0.f;
        if (r < nrow && tap_ok && c < g.src_red && n < g.src_rows)
          v[u] = j.transpose ? w[((long)c * j.Cin + g.c_lo + n) * taps + tap_src]
                             : w[((long)n * j.Cin + g.c_lo + c) * taps + tap_src];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int r = r0 + u * RP;
        if (r < nrow && tx < TR) sh[r * pitch + tx] = v[u];
      }
    }
    __syncthreads();
    const int ts0 = t0 >> tps, tsn = min(TR >> tps, g.nts - ts0);
    for (int idx = t; idx < tsn * 64; idx += 256) {
      const int lane = idx & 63, tsl = idx >> 6;
      const int i = lane & 15, gq = lane >> 4;
      const int tt = (tsl << tps) + (gq >> pls), clb = (gq & ((1 << pls) - 1)) * 8;
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = j.transpose ? (clb + e) * 16 + i : (i << cks) + clb + e;
        // (16-byte-load form: rows of `taps` floats in source tap order, nothing stored for the padded taps)
        f[e] = !vec ? sh[r * pitch + tt] : (tt < taps ? sh[r * taps + (j.transpose ? taps - 1 - tt : tt)] : 0.f);
      }
      uint4 u;
      u.x = (unsigned)f2bf(f[0]) | ((unsigned)f2bf(f[1]) << 16);
      u.y = (unsigned)f2bf(f[2]) | ((unsigned)f2bf(f[3]) << 16);
      u.z = (unsigned)f2bf(f[4]) | ((unsigned)f2bf(f[5]) << 16);
      u.w = (unsigned)f2bf(f[6]) | ((unsigned)f2bf(f[7]) << 16);
      out[(((long)(g.chunk0 + chunk) * g.nts + ts0 + tsl) * g.NT_total + g.nt0 + nt) * 64 + lane] = u;
    }
  }
}

// Second pass of a split-reduction launch: out[v, c] = alpha * act(sum_s part[s][v][c] + bias[c]), splits added in
// index order (bit-reproducible); one thread per voxel x 4 channels, coalesced on both sides.
__global__ void splitk_reduce_kernel(const float* __restrict__ part, long part_stride, int ksplit, int cpad,
                                     unsigned short* __restrict__ out, int out_ctot, int out_off, int Cout, long nvox,
                                     const float* __restrict__ bias, int act, float slope, float alpha) {
  const int c4n = Cout >> 2;
  const long total = nvox * c4n;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % c4n) * 4;
    const long v = idx / c4n;
    const float* p = part + v * cpad + c;
    float4 s = *reinterpret_cast<const float4*>(p);
    for (int k = 1; k < ksplit; ++k) {
      const float4 t = *reinterpret_cast<const float4*>(p + (long)k * part_stride);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    if (bias) { s.x += bias[c]; s.y += bias[c + 1]; s.z += bias[c + 2]; s.w += bias[c + 3]; }
    if (act) {
      s.x = s.x > 0.f ? s.x : s.x * slope; s.y = s.y > 0.f ? s.y : s.y * slope;
      s.z = s.z > 0.f ? s.z : s.z * slope; s.w = s.w > 0.f ? s.w : s.w * slope;
    }
    s.x *= alpha; s.y *= alpha; s.z *= alpha; s.w *= alpha;
    st4<BF16>(out + v * out_ctot + out_off + c, s);
  }
}

}  // namespace

int wsr_ct_splitk_reduce(const CtArgs& a, hipStream_t st) {
  const long nvox = (long)a.B * a.Xo * a.Yo * a.Zo;
  const long total = nvox * (a.Cout >> 2);
  long grid = (total + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)grid), dim3(256), 0, st, a.part, a.part_stride, a.ksplit,
                     a.NT_total * 16, (unsigned short*)a.out, a.out_ctot, a.out_off, a.Cout, nvox, a.bias, a.act,
                     a.slope, a.alpha);
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_pack_filter_frag_multi(const wsr_pack_job_t* jobs_dev, int32_t n_jobs, int32_t dtype, void* stream) {
  if (!jobs_dev || n_jobs <= 0 || n_jobs > 65535 || (dtype != WSR_BF16 && dtype != WSR_F32)) return WSR_EINVAL;
  if (dtype == WSR_F32) {
    hipLaunchKernelGGL(pack_frag_multi_gather_kernel<F32>, dim3(32, (unsigned)n_jobs), dim3(256), 0, as_stream(stream),
                       jobs_dev);
    WSR_LAUNCH_CHECK();
    return 0;
  }
  // workgroups per job: a generator's table is ~1 000 jobs of mostly 16-28 items (empty workgroups cost dispatch time),
  // a discriminator's 43 jobs of up to 512 (measured: 16 / 32 / 128 workgroups per job win at 940 / 398 / 43 jobs)
  int gx = WSR_ENV_INT("WSR_PK_GRID", n_jobs >= 512 ? 16 : (n_jobs >= 128 ? 32 : 128));
  gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
  hipLaunchKernelGGL(pack_frag_multi_kernel, dim3((unsigned)gx, (unsigned)n_jobs), dim3(256), 0, as_stream(stream), jobs_dev,
                     WSR_ENV_INT("WSR_PK_VEC", 1));
  WSR_LAUNCH_CHECK();
  return 0;
}

extern "C" int wsr_conv_tile_tpk(int32_t red_channels_padded, int32_t taps, int32_t dtype) {
  // K-step shape used by the tile kernel for a reduction over `red_channels_padded` stored channels
  return ct_tpk(red_channels_padded, taps, dtype == WSR_F32 ? 4 : 8);
}

extern "C" int64_t wsr_frag_filter_elems(int32_t rows, int32_t red, int32_t taps, int32_t dtype) {
  const int epp = dtype == WSR_F32 ? 4 : 8;
  const int redp = (red + epp - 1) / epp * epp;
  const int tpk = ct_tpk(redp, taps, epp);
  const int ck = 4 * epp / tpk;
  const long nchunks = (redp + ck - 1) / ck, nts = (taps + tpk - 1) / tpk, nt = (rows + 15) / 16;
  return nchunks * nts * nt * 64 * epp;
}

extern "C" int wsr_pack_filter_frag(const float* w, void* out, int32_t Cout, int32_t Cin, int32_t KX, int32_t KY,
                                    int32_t KZ, int32_t transpose, int32_t dtype, void* stream) {
  if (!w || !out || Cout <= 0 || Cin <= 0 || KX <= 0 || KY <= 0 || KZ <= 0) return WSR_EINVAL;
  if (dtype != WSR_BF16 && dtype != WSR_F32) return WSR_EINVAL;
  const int epp = dtype == WSR_F32 ? 4 : 8;
  const int taps = KX * KY * KZ;
  const int rows = transpose ? Cin : Cout, red = transpose ? Cout : Cin;
  const int redp = (red + epp - 1) / epp * epp;
  const int tpk = ct_tpk(redp, taps, epp);
  const int ck = 4 * epp / tpk;
  const int nchunks = (redp + ck - 1) / ck, nts = (taps + tpk - 1) / tpk, nt = (rows + 15) / 16;
  const long total = (long)nchunks * nts * nt * 64 * epp;
  long grid = (total + 255) / 256;
  if (grid > 2048) grid = 2048;
  if (dtype == WSR_F32)
    hipLaunchKernelGGL(pack_frag_kernel<F32>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), w, (float*)out, Cout,
                       Cin, KX, KY, KZ, transpose, tpk, nchunks, nts, nt);
  else
    hipLaunchKernelGGL(pack_frag_kernel<BF16>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), w,
                       (unsigned short*)out, Cout, Cin, KX, KY, KZ, transpose, tpk, nchunks, nts, nt);
  WSR_LAUNCH_CHECK();
  return 0;
}

// Shared by the forward and input-gradient entry points.  `red` = reduction channel count
// as stored (multiple of 8), gather offset in = out - (px,py,pz) + tap.
static int run_conv_tile(CtArgs& a, int red, hipStream_t st) {
  const int taps = a.KX * a.KY * a.KZ;
  const int epp = a.f32 ? 4 : 8;
  if (red % epp || a.in_ctot % epp || a.in_off % epp) return WSR_EUNSUPPORTED;
  if (a.KX > 8 || a.KY > 8 || a.KZ > 8) return WSR_EUNSUPPORTED;
  const int tpk = ct_tpk(red, taps, epp);
  const int ck = 4 * epp / tpk;
  a.nchunks = (red + ck - 1) / ck;
  a.cin_valid = red;
  if (a.in2) {  // the entry point left the first channel of the second tensor here
    if (a.in2_chunk % ck) return WSR_EUNSUPPORTED;
    a.in2_chunk /= ck;
  }
  {
    static void* zp = nullptr;
    if (!zp) {
      hipError_t e = hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero16));
      if (e != hipSuccess) return (int)e;
    }
    a.zero16 = zp;
  }
  a.vec_ok = (!a.out_planar && a.out_ctot % 4 == 0 && a.out_off % 4 == 0 &&
              (!a.res || (a.res_ctot % 4 == 0 && a.res_off % 4 == 0)))
                 ? 1
                 : 0;
  if (a.act == 2 && !(a.vec_ok && (a.Cout & 3) == 0)) return WSR_EUNSUPPORTED;  // vector epilogue only
  // (a.ws / a.ws_bytes: the split-reduction workspace of THIS call, set by the entry points from their arguments)
  if (a.ws_bytes < 0 || (a.ws == nullptr) != (a.ws_bytes == 0)) return WSR_EINVAL;
  if (a.act_c1 != 0x7FFFFFFF && (a.act_c1 & 3)) return WSR_EINVAL;
  return dispatch_ct(a, tpk, st);
}

// conv_slide.hip: sliding-window kernels of the z-folded last conv
int wsr_conv_slide_fwd(const unsigned short* in, int in_ctot, int in_off, int C, const unsigned short* wfrag, float* out,
                       int N, int B, int X, int Y, int Z, int KX, int KY, int px, int py, const float* bias,
                       const void* zero16, hipStream_t st);

int wsr_conv_slide_dgrad(const unsigned short* dy, int dy_ctot, int dy_off, int red, const unsigned short* wfrag_t,
                         unsigned short* dx, int dx_ctot, int dx_off, int C, int B, int X, int Y, int Z, int KX, int KY,
                         int px, int py, float alpha, const wsr_lrelu_mask_t* mask, const void* zero16, hipStream_t st);

static const void* zero_page() {
  static void* zp = nullptr;
  if (!zp && hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero16)) != hipSuccess) zp = nullptr;
  return zp;
}

int wsr_conv1x1_bf16(const unsigned short* in, int in_ctot, int in_off, int red, const unsigned short* wfrag,
                     unsigned short* out, int out_ctot, int out_off, int n_out, long nvox, const float* bias,
                     const unsigned short* res, int res_ctot, int res_off, int res_c1, float alpha, float beta, int act,
                     float slope, const wsr_lrelu_mask_t* mask, const unsigned short* res2, int res2_ctot, int res2_off,
                     float beta2, hipStream_t st);  // conv_1x1.hip

// conv_thin.hip: sliding-window kernel of the 3x3x3 convs with <= 16 reduction channels
int wsr_conv_thin3(const unsigned short* in, int in_ctot, int in_off, int red, const unsigned short* wfrag,
                   unsigned short* out, int out_ctot, int out_off, int n_out, int B, int X, int Y, int Z, const float* bias,
                   float alpha, int act, float slope, hipStream_t st);

// the instantiations that take a two-tensor concat (CtArgs.in2 / out2) serve this many produced channels of this dtype
static bool split_width_ok(int n, bool f32, int red, int taps, long vox) {
  if (f32) return n > 128 && n <= 144 && !WSR_ENV_SET("WSR_NO_F32_TILE");
  const int tpk = ct_tpk(red, taps, 8);
  const long nch = (red + 32 / tpk - 1) / (32 / tpk);
  return n == 144 && nch * taps >= 256 && vox >= 128L * 512;  // (dispatch_ct: smaller volumes run the 128-voxel tiles)
}

int wsr_wgrad_split_plan_ok(const wsr_conv_t* c, int c0);  // conv_wgrad.hip

extern "C" int wsr_conv_split_ok(const wsr_conv_t* c, int32_t c0) {
  if (!conv_geom_ok_split(c, c0) || WSR_ENV_SET("WSR_NO_SPLIT_CAT")) return 0;
  if ((c->sx | c->sy | c->sz) != 1 || c->upsample_xy || c->lat) return 0;
  const bool f32 = c->dtype == WSR_F32;
  const int taps = c->KX * c->KY * c->KZ, epp = f32 ? 4 : 8;
  if (taps == 1 || taps > 125 || c->KX > 8 || c->KY > 8 || c->KZ > 8) return 0;
  if (c->Cin % epp || c->Cout % epp || c->in_ctot % epp || c->in_off % epp || c->out_ctot % epp || c->out_off % epp) return 0;
  if (c0 % 128) return 0;  // whole reduction chunks / n-tiles / c-chunks of every kernel involved (16 .. 128 channels)
  const long vox_o = (long)c->B * c->Xo * c->Yo * c->Zo, vox_i = (long)c->B * c->Xi * c->Yi * c->Zi;
  // forward: produced width Cout, reduction Cin; input gradient: produced width Cin, reduction Cout
  if (!split_width_ok(c->Cout, f32, c->Cin, taps, vox_o) || !split_width_ok(c->Cin, f32, c->Cout, taps, vox_i)) return 0;
  // ... and the filter gradient: its tile planners (slots, LDS, c-chunk alignment at c0) have to accept the two-tensor
  // input as well, or the engine would find out in the middle of a backward pass
  return wsr_wgrad_split_plan_ok(c, c0) ? 1 : 0;
}

extern "C" int wsr_conv3d_fwd_tile(const wsr_conv_t* c, const void* x, const void* wfrag, void* y,
                                   const wsr_epilogue_t* ep, void* stream) {
  const bool two = ep && ep->in2;
  if (!(two ? conv_geom_ok_split(c, ep->in2_c0) : conv_geom_ok(c)) || !x || !wfrag || !y) return WSR_EINVAL;
  if (two && (ep->in2_ctot <= 0 || c->Cin - ep->in2_c0 > ep->in2_ctot || c->KX * c->KY * c->KZ == 1 || c->lat ||
              c->upsample_xy || (c->sx | c->sy | c->sz) != 1))
    return two && (ep->in2_ctot <= 0 || c->Cin - ep->in2_c0 > ep->in2_ctot) ? WSR_EINVAL : WSR_EUNSUPPORTED;
  if (c->sx > 2 || c->sy > 2 || c->sz > 2 || c->lat == 3) return WSR_EUNSUPPORTED;
  if ((c->sx | c->sy | c->sz) != 1 && (c->upsample_xy || WSR_ENV_SET("WSR_CT_NOSTRIDE"))) return WSR_EUNSUPPORTED;
  const bool f32 = c->dtype == WSR_F32;
  // fp32 (the reference's own arithmetic): stride-1 convs on the same halo-tile kernel, WSR_NO_F32_TILE=1: generic kernel
  if (f32 && ((c->sx | c->sy | c->sz) != 1 || WSR_ENV_SET("WSR_NO_F32_TILE"))) return WSR_EUNSUPPORTED;
  CtArgs a{};
  a.f32 = f32 ? 1 : 0;
  a.sx = c->sx; a.sy = c->sy; a.sz = c->sz;
  a.in = (const unsigned short*)x;
  a.wf = (const unsigned short*)wfrag;
  a.out = y;
  a.alpha = 1.f;
  if (ep) {
    a.bias = ep->bias;
    a.chan_scale = ep->chan_scale;
    a.res = (const unsigned short*)ep->res;
    a.res_ctot = ep->res_ctot;
    a.res_off = ep->res_off;
    a.res_c1 = 0x7FFFFFFF;
    a.alpha = ep->alpha;
    a.beta = ep->beta;
    a.slope = ep->slope;
    a.act = ep->act;
    a.act_c1 = ep->act_c1 > 0 ? ep->act_c1 : 0x7FFFFFFF;
    if (a.act == 2 && (!a.res || ep->out_planar)) return WSR_EINVAL;
    a.out_planar = ep->out_planar;
    if (a.res && (a.res_off < 0 || a.res_off + c->Cout > a.res_ctot)) return WSR_EINVAL;
    a.ws = ep->ws;
    a.ws_bytes = (long)ep->ws_bytes;
    if (two) {  // (in2_chunk holds the channel count until run_conv_tile knows the chunk size)
      a.in2 = (const unsigned short*)ep->in2;
      a.in2_ctot = ep->in2_ctot;
      a.in2_chunk = ep->in2_c0;
    }
    if (ep->mask) {  // LeakyReLU-backward mask on a forward-form launch (parity input gradients of strided convs)
      const wsr_lrelu_mask_t* mk = ep->mask;
      if (!mk->y || ep->out_planar || mk->c0 < 0 || mk->c1 > c->Cout || mk->c0 >= mk->c1 || mk->c0 % 4 || mk->y_ctot % 4 ||
          mk->y_off % 4 || mk->y_off + (mk->c1 - mk->c0) > mk->y_ctot || mk->chan_scale)
        return WSR_EINVAL;
      if (f32) return WSR_EUNSUPPORTED;
      a.mask_y = (const unsigned short*)mk->y;
      a.mask_ctot = mk->y_ctot; a.mask_off = mk->y_off;
      a.mask_c0 = mk->c0; a.mask_c1 = mk->c1;
      a.mask_slope = mk->slope;
    }
  }
  a.B = c->B; a.Xi = c->Xi; a.Yi = c->Yi; a.Zi = c->Zi;
  a.Xo = c->Xo; a.Yo = c->Yo; a.Zo = c->Zo;
  a.ups = c->upsample_xy ? 1 : 0;
  a.in_ctot = c->in_ctot; a.in_off = c->in_off;
  a.Cout = c->Cout; a.out_ctot = c->out_ctot; a.out_off = c->out_off;
  a.KX = c->KX; a.KY = c->KY; a.KZ = c->KZ;
  a.px = c->px; a.py = c->py; a.pz = c->pz;
  if (c->lat) {  // parity conv(s) of a sub-pixel up-sampling conv: y is written on the (2x + ox, 2y + oy) lattice
    if (a.res || a.out_planar) return WSR_EUNSUPPORTED;
    a.ol_m = 2; a.ol_ox = c->lat_ox; a.ol_oy = c->lat_oy;
    a.ol_mz = c->lat_mz > 1 ? c->lat_mz : 1; a.ol_oz = c->lat_oz;
    a.nphase = c->lat_phases == 4 ? 4 : 1;
    a.ph_wstride = (long)wsr_frag_filter_elems(c->Cout, c->Cin, c->KX * c->KY * c->KZ, c->dtype);
  }
  if (f32 && c->KX * c->KY * c->KZ == 1) return WSR_EUNSUPPORTED;  // (1x1x1 in fp32: the generic kernel)
  if (!f32 && !a.mask_y && c->KX * c->KY * c->KZ == 1 && !c->lat && !a.ups && !a.out_planar && !a.chan_scale && (c->px | c->py | c->pz) == 0 &&
      (c->sx | c->sy | c->sz) == 1 && a.act <= 1 && a.act_c1 == 0x7FFFFFFF) {
    const int rc = wsr_conv1x1_bf16(a.in, a.in_ctot, a.in_off, c->Cin, a.wf, (unsigned short*)a.out, a.out_ctot,
                                    a.out_off, a.Cout, (long)c->B * c->Xo * c->Yo * c->Zo, a.bias, a.res, a.res_ctot,
                                    a.res_off, a.res ? a.res_c1 : 0, a.alpha, a.beta, a.act, a.slope, nullptr,
                                    ep ? (const unsigned short*)ep->res2 : nullptr, ep ? ep->res2_ctot : 0,
                                    ep ? ep->res2_off : 0, ep ? ep->beta2 : 0.f, as_stream(stream));
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  if (ep && ep->res2) return WSR_EUNSUPPORTED;  // only the streaming kernel takes a second residual
  if (two) return run_conv_tile(a, c->Cin, as_stream(stream));  // (the halo-tile kernel's 9-n-tile instantiations only)
  // thin z-tapless conv with a planar fp32 result (the z-folded last conv): sliding-window kernel
  const bool no_slide = WSR_ENV_SET("WSR_NO_SLIDE");  // tuning / A-B switch
  if (!f32 && !no_slide && c->KZ == 1 && c->Cout <= 16 && a.out_planar && !a.chan_scale && !a.res && a.act == 0 && !a.ups &&
      !c->lat && (c->sx | c->sy | c->sz) == 1 && c->pz == 0 && a.alpha == 1.f && c->Xo == c->Xi && c->Yo == c->Yi &&
      zero_page()) {
    const int rc = wsr_conv_slide_fwd(a.in, a.in_ctot, a.in_off, c->Cin, a.wf, (float*)a.out, c->Cout, c->B, c->Xi, c->Yi,
                                      c->Zi, c->KX, c->KY, c->px, c->py, a.bias, zero_page(), as_stream(stream));
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  // 3x3x3 "same" conv over <= 16 stored channels (terrain convs, feature conv, the discriminator's first conv): memory-bound
  if (!f32 && (c->KX & c->KY & c->KZ) == 3 && (c->KX | c->KY | c->KZ) == 3 && (c->px & c->py & c->pz) == 1 &&
      (c->px | c->py | c->pz) == 1 && (c->sx | c->sy | c->sz) == 1 && !a.ups && !c->lat && !a.out_planar &&
      !a.chan_scale && !a.res && !a.mask_y && a.act <= 1 && a.act_c1 == 0x7FFFFFFF && c->Cin <= 16) {
    const int rc = wsr_conv_thin3(a.in, a.in_ctot, a.in_off, c->Cin, a.wf, (unsigned short*)a.out, a.out_ctot, a.out_off,
                                  a.Cout, c->B, c->Xi, c->Yi, c->Zi, a.bias, a.alpha, a.act, a.slope, as_stream(stream));
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  return run_conv_tile(a, c->Cin, as_stream(stream));
}

extern "C" int wsr_conv3d_dgrad_tile(const wsr_conv_t* c, const void* dy, const void* wfrag_t, void* dx, float alpha,
                                     int accumulate, int dx_planar, const wsr_lrelu_mask_t* mask,
                                     const wsr_dgrad_opts_t* opts, void* stream) {
  const bool two = opts && opts->dx2;
  if (!(two ? conv_geom_ok_split(c, opts->dx2_c0) : conv_geom_ok(c)) || !dy || !wfrag_t || !dx) return WSR_EINVAL;
  if (opts && opts->acc_src && !accumulate) return WSR_EINVAL;
  if (two) {
    if (opts->dx2_ctot <= 0 || c->Cin - opts->dx2_c0 > opts->dx2_ctot) return WSR_EINVAL;
    if (accumulate || dx_planar || mask || c->lat || c->upsample_xy || c->KX * c->KY * c->KZ == 1) return WSR_EUNSUPPORTED;
  }
  if (mask && (!mask->y || dx_planar || mask->c0 < 0 || mask->c1 > c->Cin || mask->c0 >= mask->c1 ||
               mask->c0 % 4 || mask->y_ctot % 4 || mask->y_off % 4 ||
               mask->y_off + (mask->c1 - mask->c0) > mask->y_ctot))
    return WSR_EINVAL;
  if ((c->sx | c->sy | c->sz) != 1 || c->lat == 3) return WSR_EUNSUPPORTED;
  const bool f32 = c->dtype == WSR_F32;
  if (f32 && WSR_ENV_SET("WSR_NO_F32_TILE")) return WSR_EUNSUPPORTED;
  const int ux = c->upsample_xy ? 2 : 1;
  CtArgs a{};
  a.f32 = f32 ? 1 : 0;
  a.in = (const unsigned short*)dy;
  a.wf = (const unsigned short*)wfrag_t;
  a.out = dx;
  a.alpha = alpha;
  a.sx = a.sy = a.sz = 1;
  a.act_c1 = 0x7FFFFFFF;
  a.out_planar = dx_planar ? 1 : 0;
  if (accumulate) {  // 1: every produced channel; n > 1: the first n (a multiple of 4) only
    if (dx_planar) return WSR_EUNSUPPORTED;
    if (accumulate > 1 && (accumulate & 3)) return WSR_EINVAL;
    a.res = (const unsigned short*)((opts && opts->acc_src) ? opts->acc_src : dx);
    a.res_ctot = c->in_ctot;
    a.res_off = c->in_off;
    a.res_c1 = accumulate > 1 ? accumulate : 0x7FFFFFFF;
    a.beta = (opts && opts->acc_beta != 0.f) ? opts->acc_beta : 1.f;
  }
  const unsigned short* res2 = opts ? (const unsigned short*)opts->res2 : nullptr;
  if (res2 && (!accumulate || opts->res2_off < 0 || opts->res2_ctot <= 0)) return WSR_EINVAL;
  if (opts) {
    a.ws = opts->ws;
    a.ws_bytes = (long)opts->ws_bytes;
  }
  a.B = c->B;
  a.Xi = c->Xo; a.Yi = c->Yo; a.Zi = c->Zo;            // gathered tensor = dy
  a.Xo = c->Xi * ux; a.Yo = c->Yi * ux; a.Zo = c->Zi;  // produced tensor = dx (fine resolution when up-sampled)
  a.ups = 0;
  a.in_ctot = c->out_ctot; a.in_off = c->out_off;
  a.Cout = c->Cin; a.out_ctot = c->in_ctot; a.out_off = c->in_off;
  a.KX = c->KX; a.KY = c->KY; a.KZ = c->KZ;
  a.px = c->KX - 1 - c->px; a.py = c->KY - 1 - c->py; a.pz = c->KZ - 1 - c->pz;
  if (c->lat) {  // parity conv of a sub-pixel up-sampling conv: dy is read on the (2x + ox, 2y + oy) lattice
    if (c->lat_phases || c->lat_mz > 1) return WSR_EUNSUPPORTED;  // the parities add into the same dx: one launch each
    a.il_m = 2; a.il_ox = c->lat_ox; a.il_oy = c->lat_oy;
  }
  if (mask && mask->chan_scale) a.chan_scale = mask->chan_scale;
  if (f32 && c->KX * c->KY * c->KZ == 1) return WSR_EUNSUPPORTED;  // (1x1x1 in fp32: the generic kernel)
  if (!f32 && c->KX * c->KY * c->KZ == 1 && !c->lat && ux == 1 && !a.out_planar && (c->px | c->py | c->pz) == 0 && !a.chan_scale) {
    const int rc = wsr_conv1x1_bf16(a.in, a.in_ctot, a.in_off, c->Cout, a.wf, (unsigned short*)a.out, a.out_ctot,
                                    a.out_off, a.Cout, (long)c->B * c->Xo * c->Yo * c->Zo, nullptr, a.res, a.res_ctot,
                                    a.res_off, a.res ? a.res_c1 : 0, a.alpha, a.beta, 0, 0.f, mask, res2,
                                    res2 ? opts->res2_ctot : 0, res2 ? opts->res2_off : 0, res2 ? opts->beta2 : 0.f,
                                    as_stream(stream));
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  if (res2) return WSR_EUNSUPPORTED;  // only the streaming kernel takes a second residual
  if (two) {  // the gradient of a two-tensor concat: channels >= dx2_c0 leave for the second tensor
    a.out2 = opts->dx2;
    a.out2_ctot = opts->dx2_ctot;
    a.out2_c0 = opts->dx2_c0;
    return run_conv_tile(a, c->Cout, as_stream(stream));
  }
  // z-tapless conv with a thin output side and the mask of the layer below (the z-folded last conv): sliding window
  const bool no_slide = WSR_ENV_SET("WSR_NO_SLIDE");  // tuning / A-B switch
  if (!f32 && !no_slide && mask && c->KZ == 1 && c->Cout <= 16 && !accumulate && !dx_planar && !c->lat && ux == 1 && c->pz == 0 &&
      c->Xo == c->Xi && c->Yo == c->Yi && c->Zo == c->Zi && zero_page()) {
    const int rc = wsr_conv_slide_dgrad(a.in, a.in_ctot, a.in_off, c->Cout, a.wf, (unsigned short*)a.out, a.out_ctot,
                                        a.out_off, c->Cin, c->B, c->Xi, c->Yi, c->Zi, c->KX, c->KY, a.px, a.py, alpha, mask,
                                        zero_page(), as_stream(stream));
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  if (!f32 && !mask && !accumulate && !dx_planar && !c->lat && ux == 1 && (c->KX & c->KY & c->KZ) == 3 &&
      (c->KX | c->KY | c->KZ) == 3 && (c->px & c->py & c->pz) == 1 && (c->px | c->py | c->pz) == 1 && c->Cout <= 16) {
    // input gradient of such a conv with a thin OUTPUT side (terrain_convs.1, 16 -> 16): the same sliding-window kernel
    const int rc = wsr_conv_thin3(a.in, a.in_ctot, a.in_off, c->Cout, a.wf, (unsigned short*)a.out, a.out_ctot, a.out_off,
                                  a.Cout, c->B, c->Xi, c->Yi, c->Zi, nullptr, alpha, 0, 0.f, as_stream(stream));
    if (rc != WSR_EUNSUPPORTED) return rc;
  }
  if (mask) {
    a.mask_y = (const unsigned short*)mask->y;
    a.mask_ctot = mask->y_ctot; a.mask_off = mask->y_off;
    a.mask_c0 = mask->c0; a.mask_c1 = mask->c1;
    a.mask_slope = mask->slope;
  }
  return run_conv_tile(a, c->Cout, as_stream(stream));
}
