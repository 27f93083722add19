/*
 * windsr_hip.h - C ABI of the MI355X (gfx950) kernels behind the 3D-conv GAN
 * train-step hot path of GAN_SR_wind_field.
 *
 * The reference has no FFI layer for this path: its convolutions execute inside
 * PyTorch/ATen.  Each entry point below names the reference call site (file:line
 * under the reference tree) whose ATen work it replaces.  All pointers are DEVICE
 * pointers unless stated otherwise; the caller owns every buffer; kernels never
 * allocate, never synchronise and run on the hipStream_t passed as `stream`
 * (NULL = the default stream).  Every function returns 0 on success, a negative
 * WSR_E* code for bad arguments or the positive hipError_t of a failed launch.
 *
 * Activation layout ("NDHWC"): element (b, x, y, z, c) of a tensor with `ctot`
 * channels per voxel lives at ((((b*X + x)*Y + y)*Z + z)*ctot + c); a conv may
 * read/write a channel window [off, off+C) of such a buffer, which is how the
 * dense-block concatenations (torch_blocks.py:212-214, Generator_3D...py:228)
 * are done without a copy.  "Planar" tensors are the reference's own fp32
 * (B, C, X, Y, Z) contiguous tensors (network inputs / outputs).
 * Packed (compute) filter layout: [Cout][KX][KY][KZ][Cin] (Cin fastest), made by
 * wsr_pack_filter from the master weights, which keep nn.Conv3d's own logical
 * layout (Cout, Cin, KX, KY, KZ) so checkpoints / init / Adam are untouched.
 */
#ifndef WINDSR_HIP_H
#define WINDSR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WSR_ABI_VERSION 9

enum wsr_dtype { WSR_F32 = 0, WSR_BF16 = 1 };

enum wsr_error {
  WSR_OK = 0,
  WSR_EINVAL = -1,      /* inconsistent geometry / null pointer            */
  WSR_EUNSUPPORTED = -2 /* shape outside what the kernels were built for   */
};

/* Geometry of one 3-D convolution.  X/Y/Z naming follows the reference
 * (tensor dims 2,3,4; Z = vertical levels, contiguous, never up-scaled). */
typedef struct wsr_conv {
  int32_t dtype;                 /* wsr_dtype of activations and packed filters */
  int32_t B;                     /* batch                                         */
  int32_t Xi, Yi, Zi;            /* stored input extent                           */
  int32_t Xo, Yo, Zo;            /* output extent                                 */
  int32_t Cin, in_ctot, in_off;  /* input channel window                          */
  int32_t Cout, out_ctot, out_off;
  int32_t KX, KY, KZ;
  int32_t sx, sy, sz;            /* stride                                        */
  int32_t px, py, pz;            /* zero padding                                  */
  int32_t upsample_xy;           /* 1: input is read through nearest x(2,2,1)
                                    up-sampling (torch_blocks.py:345-347)        */
  /* Sub-pixel form of that up-sampling conv (tile entry points only; all zero = none).  Nearest x(2,2,1)
   * followed by a 3x3xKZ conv equals four 2x2xKZ convs on the un-sampled input, one per output parity (a, b):
   * 4/9 of the multiply-adds (filters from wsr_subpixel_fold).  lat = 2 marks such a parity conv: a same-size
   * conv (Xo = Xi, Yo = Yi, stride 1, low pads px / py, high pads K-1-p) whose OUTPUT voxel (x, y, z) is voxel
   * (2x + lat_ox, 2y + lat_oy, z) of a tensor of extent (2Xo, 2Yo, Zo) - y in the forward pass, dy in the
   * input / filter gradient.  lat_phases = 4 (wsr_conv3d_fwd_tile only): all four parities in ONE launch -
   * parity (a, b) runs with low pads (px - a, py - b), lattice offsets (a, b) and the fragment filter at
   * wfrag + (2a + b) * wsr_frag_filter_elems(Cout, Cin, taps).
   * The same machinery runs the INPUT gradient of the discriminator's stride-(2,2,s) 4x4x3 convs
   * (torch_blocks.py:372-521) as forward convs over dy: input voxel (2m + a, 2n + b, s*l + c) only meets the filter
   * taps of matching parity, so each parity class is a 2x2xKZ' conv on dy that writes its own lattice of dx (filters
   * from wsr_strided_parity_filters).  lat_mz = 2 puts the output on the z lattice lat_mz*z + lat_oz as well
   * (0 / 1: z is not strided); with lat set the conv is same-size along z too (Zo = Zi, low pad pz).
   * lat = 3 (filter-gradient entry points only) is the mirror image for the FILTER gradient of those down-sampling
   * convs: filter tap (2i + a, 2j + b, .) only meets input voxels of one parity, so the gradient of the taps of a
   * class is a stride-1 2x2xKZ' filter gradient in which the INPUT x is read on the lattice (2x + lat_ox,
   * 2y + lat_oy, lat_mz*z + lat_oz) of the stored tensor (Xi, Yi, Zi = the lattice's extents = Xo, Yo, Zo) and dy
   * is dense; wsr_strided_parity_unfold moves the class gradients to their taps of the master gradient.     */
  int32_t lat, lat_ox, lat_oy, lat_phases;
  int32_t lat_mz, lat_oz;
} wsr_conv_t;

struct wsr_lrelu_mask;
/* Fused epilogue:  v = conv (+ bias[c]);  v = lrelu(v, slope) if act;
 *                  v *= chan_scale[b*Cout + c] if chan_scale (Dropout3d mask);
 *                  y = alpha*v (+ beta*res[...,res_off + c] if res).          */
typedef struct wsr_epilogue {
  const float* bias;        /* [Cout] or NULL                                   */
  const float* chan_scale;  /* [B*Cout] or NULL                                 */
  const void* res;          /* NDHWC tensor of `dtype`, or NULL                 */
  int32_t res_ctot, res_off;
  float alpha, beta;
  int32_t act;              /* 0 none, 1 leaky-relu, 2 leaky-relu AFTER the residual: y = alpha*lrelu(acc + bias +
                               beta*res) (tile kernels only) - the second stage of a split dense-block conv whose
                               partial sums over the block input are already in `res` (= the output window) */
  float slope;
  int32_t out_planar;       /* 1: y is fp32 planar (B, Cout, Xo, Yo, Zo)        */
  int32_t act_c1;           /* > 0 (tile kernels only): bias and activation apply to channels < act_c1 only, the rest
                               are stored as raw sums - the first stage of a split dense-block conv            */
  /* ABI 6 - split-reduction workspace of THIS call (tile entry point only; NULL / 0 = never split): device memory the
   * caller owns.  With it, launches that would otherwise run on a few workgroups with a long reduction (the deep
   * layers of the discriminator, Discriminator_3D.py:66-169: 128..1024 voxels x 256..512 channels x 27..48 taps)
   * split the reduction channels over up to 256 workgroups; the fp32 partial sums go through the workspace and a
   * second launch on the same stream adds them in index order (bit-reproducible) and applies the epilogue.  Launches
   * that may run concurrently (different streams / threads) take different workspaces - the library keeps no
   * state between calls.  Results do not depend on its presence beyond fp32 summation order.                 */
  void* ws;
  int64_t ws_bytes;
  const void* res2;         /* second residual, y += beta2*res2[..., res2_off + c] (streaming 1x1x1 kernel only,
                               WSR_EUNSUPPORTED elsewhere): the last LFF of an RRDB adds the dense block's and the
                               RRDB's shortcut at once (torch_blocks.py:290,330)                                */
  int32_t res2_ctot, res2_off;
  float beta2;
  /* ABI 6 (tile entry point only, NULL = none): the LeakyReLU-backward mask of wsr_conv3d_dgrad_tile on a FORWARD-form
   * launch - the parity input gradients of the discriminator's strided convs are forward convs over dy (wsr_conv_t.lat):
   * with the mask of the layer below (Discriminator_3D.py:67-75: conv + LeakyReLU, no BatchNorm) in their epilogue
   * that layer's leaky_relu_backward needs no pass of its own.  bf16, <= 32 or 65..224 produced channels
   * (WSR_EUNSUPPORTED otherwise).                                                                           */
  const struct wsr_lrelu_mask* mask;
  /* ABI 8 (tile entry point only, NULL = none): the conv's input is the channel concatenation of TWO tensors
   * (torch.cat((x, Zf), 1) of Generator_3D_Resnet_ESRGAN.py:228 in front of hr_convs[0]) WITHOUT the concatenated copy
   * and without one producer writing 32-byte pieces into the other's 288-byte voxel rows (partial cache lines: 1.6 x
   * the bytes at the memory side): reduction channels [0, in2_c0) of the conv are read from `x` (window in_off of
   * in_ctot channels, as always), channels [in2_c0, Cin) from channels [0, Cin - in2_c0) of `in2` (same extents,
   * in2_ctot channels per voxel, same dtype).  in2_c0 must be a multiple of the kernel's reduction chunk (16 bf16 /
   * 8 fp32 channels).  Only the 512-voxel 129..144-output tile instantiations (the 5x5x5 144 -> 144 conv) carry it:
   * WSR_EUNSUPPORTED elsewhere - ask wsr_conv_split_ok first and keep the concatenated buffer when it says no. */
  const void* in2;
  int32_t in2_ctot, in2_c0;
} wsr_epilogue_t;

/* 1 when wsr_conv3d_fwd_tile (wsr_epilogue_t.in2), wsr_conv3d_dgrad_tile (wsr_dgrad_opts_t.dx2) and
 * wsr_conv3d_wgrad_parts_x2 all take this conv with its input channels split at `c0` into two tensors, else 0
 * (host-side query, no launch).                                                                              */
int wsr_conv_split_ok(const wsr_conv_t* c, int32_t c0);

int wsr_abi_version(void);
const char* wsr_error_string(int code);
/* The WSR_* environment switches (tuning / A-B aids, none needed for normal use) are cached per call site; a process
 * that changes its environment at run time calls this to have them read again.  Always returns 0. */
int wsr_reload_env(void);

/* ---- convolution ------------------------------------------------------------
 * aten::conv3d forward of nn.Conv3d (torch_blocks.py:17,278; Generator_3D...py:105)
 * + the LeakyReLU / cat / residual / Dropout3d / Upsample ops fused around it
 * (torch_blocks.py:35,214,290,330,46,347; Generator_3D...py:104,228).          */
int wsr_conv3d_fwd(const wsr_conv_t* c, const void* x, const void* w, void* y,
                   const wsr_epilogue_t* ep, void* stream);

/* aten::convolution_backward, input gradient.  `wt` is the filter re-packed as
 * [Cin][KX][KY][KZ][Cout] (wsr_pack_filter with transpose=1).  dx gets channel
 * window [in_off, in_off+Cin) of an `in_ctot` buffer at the conv's *stored*
 * input resolution unless upsample_xy, in which case dx is at 2Xi x 2Yi and
 * wsr_upsample2_bwd folds it.  dx = alpha*result (+ dx if accumulate; the tile entry point also takes
 * accumulate = n > 1: only the first n produced channels are added to).  dx_planar=1: dx
 * is an fp32 planar (B, Cin, X, Y, Z) tensor (gradient w.r.t. a network input).  */
int wsr_conv3d_dgrad(const wsr_conv_t* c, const void* dy, const void* wt, void* dx, float alpha,
                     int accumulate, int dx_planar, void* stream);

/* ---- LDS halo-tile path (bf16, stride 1) ------------------------------------
 * (bf16; fp32 as well for stride-1 convs - ABI 6: the same kernel on 16-byte pieces of 4 channels, exact fp32
 * products and sums through v_mfma_f32_16x16x4_f32; 1x1x1 and strided fp32 convs return WSR_EUNSUPPORTED.)
 * Same contracts as wsr_conv3d_fwd / wsr_conv3d_dgrad, but the activation tile
 * (with halo) is staged in LDS once per channel chunk and re-used by every tap,
 * and the filter comes in MFMA-fragment order from wsr_pack_filter_frag
 * (transpose = 0 for the forward pass, 1 for the input gradient; element count
 * from wsr_frag_filter_elems(rows, reduction channels, taps)).  Return
 * WSR_EUNSUPPORTED for shapes outside the tile kernels (strided convs, fp32):
 * the caller then uses the generic entry points above.                          */
int wsr_conv3d_fwd_tile(const wsr_conv_t* c, const void* x, const void* wfrag, void* y,
                        const wsr_epilogue_t* ep, void* stream);
/* leaky_relu_backward folded into the input-gradient epilogue (optional, NULL = none): after the
 * accumulation, produced channels [c0, c1) are multiplied by (y > 0 ? 1 : slope), y = channel
 * y_off + (c - c0) of the saved forward output at the same voxel (NDHWC bf16, y_ctot channels).  This is
 * the LeakyReLU of the layer whose OUTPUT gradient those channels are (dense-block growth channels). */
typedef struct wsr_lrelu_mask {
  const void* y;
  int32_t y_ctot, y_off, c0, c1;
  float slope;
  const float* chan_scale;  /* optional [B][Cin]: every produced channel is also multiplied by it - the
                               Dropout3d keep factors of that layer (Generator_3D...py:104), backward       */
} wsr_lrelu_mask_t;
/* ABI 6 - optional extras of the tile input gradient (NULL = none of them):
 *   acc_src : with `accumulate`, the tensor whose values are added (dx's own layout: in_ctot channels per voxel, the
 *             window at in_off) - NULL = dx itself.  Lets the running gradient of a residual chain move from one
 *             buffer to the next instead of being updated in place, so that a filter-gradient launch on another
 *             stream may still read the previous buffer (the dense blocks' backward pass, torch_blocks.py:256-290).
 *   ws / ws_bytes : split-reduction workspace of this call, as wsr_epilogue_t.ws.                            */
typedef struct wsr_dgrad_opts {
  const void* acc_src;
  void* ws;
  int64_t ws_bytes;
  /* The running gradient of a residual-in-residual block (torch_blocks.py:293-330: out = x + scale * chain(x)) without
   * copy / add passes - streaming 1x1x1 kernel only (WSR_EUNSUPPORTED elsewhere when res2 is set):
   *   acc_beta : weight of the accumulated values (0 = 1): dx = alpha * conv^T(dy) + acc_beta * acc_src (first n channels);
   *   res2     : a second tensor added to the same first n channels, dx += beta2 * res2[..., res2_off + c] - the gradient
   *              of the RRDB's output, which joins the chain's gradient where the chain ends.                     */
  float acc_beta;
  float beta2;
  const void* res2;
  int32_t res2_ctot, res2_off;
  /* ABI 8 - the mirror image of wsr_epilogue_t.in2: the input gradient of a conv over a two-tensor concatenation is
   * written to TWO tensors - produced channels [0, dx2_c0) to dx (window in_off of in_ctot, as always), channels
   * [dx2_c0, Cin) to channels [0, Cin - dx2_c0) of `dx2` (dx2_ctot channels per voxel).  No accumulate / mask /
   * planar output with it; same instantiations as in2 (WSR_EUNSUPPORTED elsewhere).                         */
  void* dx2;
  int32_t dx2_ctot, dx2_c0;
} wsr_dgrad_opts_t;
int wsr_conv3d_dgrad_tile(const wsr_conv_t* c, const void* dy, const void* wfrag_t, void* dx, float alpha,
                          int accumulate, int dx_planar, const wsr_lrelu_mask_t* mask, const wsr_dgrad_opts_t* opts,
                          void* stream);
/* ABI 6: `dtype` (wsr_dtype) of the fragment filter - bf16, or fp32 for the fp32 tile kernels (stride-1 convs in the
 * reference's own arithmetic; the 16-byte pieces then hold 4 channels instead of 8).                        */
int64_t wsr_frag_filter_elems(int32_t rows, int32_t red, int32_t taps, int32_t dtype);
int wsr_pack_filter_frag(const float* w, void* out, int32_t Cout, int32_t Cin, int32_t KX, int32_t KY, int32_t KZ,
                         int32_t transpose, int32_t dtype, void* stream);
/* The same for many filters in ONE launch (every filter changes at each optimizer step, and a
 * generator has ~300 of them): `jobs_dev` is a DEVICE array of n_jobs records.                    */
typedef struct wsr_pack_job {
  const float* w;  /* master (Cout, Cin, KX, KY, KZ) fp32 */
  void* out;       /* wsr_frag_filter_elems(...) bf16 elements */
  int32_t Cout, Cin, KX, KY, KZ, transpose;
  /* Sub-block of a stacked filter of a residual dense block (torch_blocks.py:256-267), red_total > 0: the
   * destination is the filter of ONE virtual conv with rows_total rows and red_total reduction channels,
   * several jobs fill disjoint parts of it.
   *   transpose = 1 (stacked input gradient): this conv's output channels are reduction channels
   *     [red_off, red_off + Cout) and its input channels [c_lo, c_lo + c_n) the rows (rows_total = c_n);
   *   transpose = 0 (split forward conv): this conv's output channels are rows [row_off, row_off + Cout) and
   *     its input channels [c_lo, c_lo + c_n) the whole reduction axis (red_total = c_n, red_off = 0).
   * Offsets and extents that index whole chunks / n-tiles must be multiples of 16 (fp32: 8).  red_total = 0: a plain
   * filter (the fields above).                                                                        */
  int32_t c_lo, c_n, red_off, red_total, row_off, rows_total;
} wsr_pack_job_t;
int wsr_pack_filter_frag_multi(const wsr_pack_job_t* jobs_dev, int32_t n_jobs, int32_t dtype, void* stream);

/* aten::convolution_backward, filter gradient: dw[Cout][taps][Cin] fp32 (packed
 * order) is ACCUMULATED into (caller zeroes it when a fresh gradient is wanted;
 * wsr_unpack_wgrad moves it to the master layout).                             */
int wsr_conv3d_wgrad(const wsr_conv_t* c, const void* x, const void* dy, float* dw, void* stream);
/* Deterministic (atomic-free, bit-reproducible) form of the same gradient.  The kernels split the reduction
 * over the voxels between workgroups / waves; wsr_conv3d_wgrad adds the partial sums into one buffer with float
 * atomics (order varies from launch to launch), here split s STORES its sums to parts + s*part_stride (each a
 * packed [Cout][taps][Cin] image; nothing has to be zeroed) and wsr_unpack_wgrad_reduce_multi adds the copies
 * in index order while moving them to the master layout.  n_parts must be the value wsr_conv3d_wgrad_nparts
 * reports for the same geometry (host-side query, no launch); tri_base / tri_step > 0: the stacked dense-block
 * form of wsr_conv3d_wgrad_tri.                                                                             */
int wsr_conv3d_wgrad_nparts(const wsr_conv_t* c, int32_t tri_base, int32_t tri_step, int32_t* n_parts);
int wsr_conv3d_wgrad_parts(const wsr_conv_t* c, const void* x, const void* dy, float* parts, int64_t part_stride,
                           int32_t n_parts, int32_t tri_base, int32_t tri_step, void* stream);
/* ABI 8 - the same with the conv's input split over two tensors (wsr_epilogue_t.in2): input channels [x2_c0, Cin)
 * are read from channels [0, Cin - x2_c0) of `x2` (x2_ctot channels per voxel); x2_c0 a multiple of 32.  The
 * packed [Cout][taps][Cin] image is the un-split conv's.  bf16 / fp32 tile filter-gradient kernels, stride 1, no
 * lattice (WSR_EUNSUPPORTED otherwise).                                                                        */
int wsr_conv3d_wgrad_parts_x2(const wsr_conv_t* c, const void* x, const void* x2, int32_t x2_ctot, int32_t x2_c0,
                              const void* dy, float* parts, int64_t part_stride, int32_t n_parts, void* stream);
/* Filter gradients of ALL growth convs of a residual dense block in one launch
 * (torch_blocks.py:256-267: conv i reads channels [0, tri_base + i*tri_step) of the
 * dense buffer and writes tri_step channels).  c describes the stacked conv: Cin =
 * the widest input window, Cout = n_convs*tri_step, dy = the stacked output
 * gradients.  Row n of dw belongs to conv n / tri_step; its entries with
 * c >= tri_base + (n / tri_step)*tri_step are unspecified.  bf16, stride 1 only
 * (WSR_EUNSUPPORTED otherwise).                                                 */
int wsr_conv3d_wgrad_tri(const wsr_conv_t* c, const void* x, const void* dy, float* dw, int32_t tri_base,
                         int32_t tri_step, void* stream);

/* ---- filter packing ----------------------------------------------------------
 * master fp32 (Cout, Cin, KX, KY, KZ) contiguous -> compute copy of `dtype`:
 *   transpose=0: [Cout][taps][kpad]  (channels >= Cin zero-filled)
 *   transpose=1: [Cin][taps][kpad]   (channels >= Cout zero-filled; dgrad operand)
 * kpad lets 1/3/4-channel tensors (LR fields, terrain height, D input, SR output:
 * Generator_3D...py:78-85,105-110,120-127; Discriminator_3D.py:67) use the
 * 16-byte-piece MFMA path.                                                     */
int wsr_pack_filter(const float* w, void* out, int32_t dtype, int32_t Cout, int32_t taps,
                    int32_t Cin, int32_t transpose, int32_t kpad, void* stream);
/* Filter gradient back to the master layout:
 * dst (Cout, Cin, taps) = scale * src [Cout][taps][kpad] (+ dst if accumulate).  */
int wsr_unpack_wgrad(const float* src, float* dst, int32_t Cout, int32_t taps, int32_t Cin,
                     int32_t kpad, float scale, int32_t accumulate, void* stream);

/* The same for many filter gradients in ONE launch; `jobs_dev` is a DEVICE array of n_jobs records. */
typedef struct wsr_unpack_job {
  const float* src; /* packed [Cout][taps][kpad] */
  float* dst;       /* master (Cout, Cin, taps)   */
  int32_t Cout, taps, Cin, kpad;
  float scale;
  int32_t accumulate;
  int32_t n_parts;     /* wsr_unpack_wgrad_reduce_multi: split copies to sum (src + s*part_stride), else unused */
  int32_t reserved;
  int64_t part_stride; /* elements between two copies */
} wsr_unpack_job_t;
int wsr_unpack_wgrad_multi(const wsr_unpack_job_t* jobs_dev, int32_t n_jobs, void* stream);
/* The same with the ordered sum over the split copies of wsr_conv3d_wgrad_parts folded in (taps <= 128). */
int wsr_unpack_wgrad_reduce_multi(const wsr_unpack_job_t* jobs_dev, int32_t n_jobs, void* stream);

/* ---- elementwise / normalisation ---------------------------------------------
 * leaky_relu_backward from the saved OUTPUT sign, in place on a channel window
 * (torch_blocks.py:35 autograd):  g *= (y > 0 ? 1 : slope), optionally also
 * times the Dropout3d keep factor chan_scale[b*C + c] (Generator_3D...py:104).  */
int wsr_lrelu_bwd_inplace(void* g, int32_t g_ctot, int32_t g_off, const void* y, int32_t y_ctot,
                          int32_t y_off, int32_t C, int64_t nvox, float slope,
                          const float* chan_scale, int64_t vox_per_b, int32_t dtype, void* stream);
/* copy / axpby on channel windows: dst = alpha*src (+ beta*dst)                */
int wsr_chan_axpby(void* dst, int32_t d_ctot, int32_t d_off, const void* src, int32_t s_ctot,
                   int32_t s_off, int32_t C, int64_t nvox, float alpha, float beta, int32_t dtype,
                   void* stream);
/* Bias gradient of a conv (aten::convolution_backward's third output; the LFF bias of a residual dense
 * block, torch_blocks.py:278): out[c] = scale * sum over voxels of x[v, x_off + c], fp32.  `out` is
 * overwritten.  Two passes (per-workgroup partial rows, then a column sum: float atomics on C addresses
 * serialise) through `partials`, a caller-owned scratch of WSR_CHAN_SUM_ROWS * C floats.  C must be a
 * multiple of 4 and <= 1024 (WSR_EUNSUPPORTED otherwise).                                          */
#define WSR_CHAN_SUM_ROWS 512
int wsr_chan_sum(const void* x, int32_t x_ctot, int32_t x_off, int32_t C, int64_t nvox, float scale, float* out,
                 float* partials, int32_t dtype, void* stream);
/* First pass of wsr_chan_sum alone: row r of `partials` (wsr_chan_sum_rows(C, nvox) rows of C floats, caller-owned)
 * receives the sums of workgroup r.  The rows can then be added by wsr_unpack_wgrad_reduce_multi together with the
 * filter gradients of the same backward pass (job: Cout = 1, taps = 1, Cin = kpad = C, n_parts = rows, part_stride =
 * C) - one launch less per bias gradient (48 per backward pass of the generator).                          */
int wsr_chan_sum_rows(int32_t C, int64_t nvox);
int wsr_chan_sum_partials(const void* x, int32_t x_ctot, int32_t x_off, int32_t C, int64_t nvox, float* partials,
                          int32_t dtype, void* stream);
/* backward of nearest x(2,2,1) up-sampling: dx[b,x,y,z,c] = sum of the 4 dy    */
int wsr_upsample2_bwd(const void* dy, void* dx, int32_t B, int32_t Xi, int32_t Yi, int32_t Zi,
                      int32_t C, int32_t dtype, void* stream);
/* Filters of the sub-pixel form of an up-sampling conv (wsr_conv_t.lat): master fp32 w (n, 3, 3, KZ), n =
 * Cout*Cin, -> wp (4, n, 2, 2, KZ), parity (a, b) at index 2a + b:
 *   wp[2a+b][f][i][j][kz] = sum over kx in S_a(i), ky in S_b(j) of w[f][kx][ky][kz],
 *   S_0(0) = {0}, S_0(1) = {1, 2} (input offsets -1, 0);  S_1(0) = {0, 1}, S_1(1) = {2} (offsets 0, +1)
 * - the taps of the 3x3 filter that read the same un-sampled voxel, summed in fp32 (torch_blocks.py:345-347:
 * nn.Upsample(scale_factor=(2,2,1), mode="nearest") in front of the conv).  _unfold is the adjoint:
 * dw[f][kx][ky][kz] = sum over (a, b, i, j) with kx in S_a(i), ky in S_b(j) of dwp[2a+b][f][i][j][kz]
 * (dw is overwritten).                                                                                  */
/* Parity filters of the input gradient of a stride-(2, 2, sz) conv with a 4x4x3 filter and padding 1 (the
 * discriminator's down-sampling convs): w (Cout, Cin, 4, 4, 3) fp32 -> out (4, Cin, Cout, 2, 2, KZp), parity (a, b)
 * at index 2a + b, for z parity class zc:
 *   out[2a+b][ci][co][i][j][t] = w[co][ci][kx(a,i)][ky(b,j)][kz(t)],  k(0,.) = {3, 1} (dy offsets -1, 0),
 *   k(1,.) = {2, 0} (offsets 0, +1);  sz = 1: KZp = 3, kz(t) = 2 - t (offsets -1, 0, +1);  sz = 2: zc = 0: KZp = 1,
 *   kz = 1 (offset 0); zc = 1: KZp = 2, kz = {2, 0} (offsets 0, +1).
 * Rows are the conv's INPUT channels: the result is the filter of a forward conv over dy (wsr_conv_t.lat).   */
int wsr_strided_parity_filters(const float* w, float* out, int32_t Cout, int32_t Cin, int32_t sz, int32_t zc,
                               void* stream);
/* Filter gradient of such a conv in parity form (wsr_conv_t.lat = 3): class (a, b) of the taps (2i + a, 2j + b) is a
 * (2,2,KZp) stride-1 filter gradient over the input sub-lattice (1 - a, 1 - b[, z class]) with low pads (1 - a, 1 - b);
 * sz = 1: KZp = 3, kz = kk, pad 1;  sz = 2: zc = 0: KZp = 1, kz = 1, z lattice offset 0, pad 0;  zc = 1: KZp = 2,
 * kz = 2 kk, z lattice offset 1, pad 1.  dwp (4, n = Cout*Cin, 2, 2, KZp) fp32, class (a, b) at index 2a + b ->
 * dw[f][2i + a][2j + b][kz] of the master gradient (n, 4, 4, 3); every element is written by exactly one class.   */
int wsr_strided_parity_unfold(const float* dwp, float* dw, int64_t n, int32_t sz, int32_t zc, void* stream);
int wsr_subpixel_fold(const float* w, float* wp, int64_t n, int32_t KZ, void* stream);
int wsr_subpixel_unfold(const float* dwp, float* dw, int64_t n, int32_t KZ, void* stream);
/* planar fp32 (B,C,X,Y,Z) <-> NDHWC `dtype` window; c_fill >= C channels are
 * written, those beyond C with zeros                                           */
int wsr_planar_to_ndhwc(const float* src, void* dst, int32_t B, int32_t C, int64_t vox_per_b,
                        int32_t d_ctot, int32_t d_off, int32_t c_fill, int32_t dtype, void* stream);
int wsr_ndhwc_to_planar(const void* src, float* dst, int32_t B, int32_t C, int64_t vox_per_b,
                        int32_t s_ctot, int32_t s_off, int32_t dtype, void* stream);

/* Wind-field derivatives of the physics losses (process_data.py:273-313 of the reference:
 * calculate_gradient_of_wind_field = torch.gradient over x, y with coordinate spacing + calculate_div_z on
 * the terrain-following levels): out[b, 3*a + c] = d f[b, c] / d axis_a, a = 0 (x), 1 (y), 2 (z), planar
 * fp32 (B, ., X, Y, Z).  Interior points use the second-order three-point stencil for non-uniform
 * spacing, the two ends one-sided first differences.  xs[X], ys[Y] are the horizontal coordinates, zc
 * (B, 1, X, Y, Z) the height of every level.  _bwd applies the adjoint (transpose) of that linear map to
 * g (B, 9, X, Y, Z): the gradient w.r.t. f.                                                        */
int wsr_wind_gradient(const float* f, const float* xs, const float* ys, const float* zc, float* out, int32_t B,
                      int32_t X, int32_t Y, int32_t Z, void* stream);
int wsr_wind_gradient_bwd(const float* g, const float* xs, const float* ys, const float* zc, float* df, int32_t B,
                          int32_t X, int32_t Y, int32_t Z, void* stream);

/* nn.Linear forward for a handful of rows and a very long reduction (the discriminator's first classifier layer,
 * Discriminator_3D.py:171-175: 1..8 samples x 100 outputs x 256*4*4*z features): y (B, N) = x (B, K) w^T (N, K) +
 * bias, fp32, one workgroup per output row streaming its weight row once -> aten::addmm.  B <= 8 and K a multiple
 * of 4 (WSR_EUNSUPPORTED otherwise: the caller uses the library GEMM).  Bit-reproducible.                  */
int wsr_linear_rows(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t N, int64_t K,
                    void* stream);
/* out[c] = sum over b, v of src[b][c][v] for a planar fp32 (B, C, V) tensor - the bias gradient of a conv whose
 * output gradient arrives planar (hr_convs.2, Generator_3D_Resnet_ESRGAN.py:105-110; aten: sum.dim_IntList).
 * Two passes through `partials` (WSR_CHAN_SUM_ROWS * C floats), no atomics.                                  */
int wsr_plane_sum(const float* src, int32_t B, int32_t C, int64_t V, float* out, float* partials, void* stream);

/* Fused content losses of the generator (reference GAN_models/wind_field_GAN_3D.py:377-432: pixel L1 / L2,
 * xy-gradient, z-gradient, divergence and xy-divergence MSE terms over the Jacobians of
 * calculate_gradient_of_wind_field, normalised by get_norm_factors_of_gradients :773-814).  Every normaliser is
 * a scalar, so one pass over hr, sr (B,3,X,Y,Z) and zc (B,1,X,Y,Z), planar fp32, yields everything:
 *   stats[0..5]  = sum (J_sr - J_hr)^2 over channels 0..5 | over channels 6..8 | sum (div3_hr - div3_sr)^2 |
 *                  sum (div2_hr - div2_sr)^2 | sum |hr - sr| | sum (hr - sr)^2
 *   stats[6..13] = max |J[:6]|, max J[6:] (signed, as the reference), max |div3|, max |div2| of hr, then of sr
 * (maxima propagate NaN like torch.max).  workspace: wsr_physics_loss_workspace_floats() floats; the reduction
 * is two-pass and atomic-free (bit-reproducible).  _bwd: coef[6] (DEVICE array) = d loss / d stats[0..5];
 * residual = scratch of B*9*X*Y*Z floats; dsr (B,3,X,Y,Z) = d loss / d sr (overwritten).                  */
int64_t wsr_physics_loss_workspace_floats(void);
int wsr_physics_loss_stats(const float* hr, const float* sr, const float* xs, const float* ys, const float* zc,
                           float* stats, float* workspace, int32_t B, int32_t X, int32_t Y, int32_t Z, void* stream);
int wsr_physics_loss_bwd(const float* hr, const float* sr, const float* xs, const float* ys, const float* zc,
                         const float* coef, float* residual, float* dsr, int32_t B, int32_t X, int32_t Y, int32_t Z,
                         void* stream);

/* z-folded form of a conv with very few output channels (the last conv of the generator, 144 -> 3,
 * 5x5x5, Generator_3D_Resnet_ESRGAN.py:120-127): the KZ taps along z become output channels of a
 * (KX,KY,1) conv with C*KZ outputs - 15 instead of 3 of the 16 columns of an MFMA tile do work, a fifth
 * of the K-steps - and the z taps are summed afterwards:
 *   fold  : y[b,c,p,z]        = bias[c] + sum_kz t[b, c*KZ+kz, p, z + kz - pz]      (planar fp32, p = x*Y+y)
 *   unfold: d[b,p,z, c*KZ+kz] = g[b,c,p, z - kz + pz]   (its adjoint: planar fp32 -> NDHWC `dtype`;
 *           channels [C*KZ, c_fill) are written as zeros); out-of-range z contributes 0.          */
int wsr_zfold(const float* t, float* y, const float* bias, int32_t B, int32_t C, int32_t KZ, int32_t pz,
              int64_t planes, int32_t Z, void* stream);
int wsr_zunfold(const float* g, void* d, int32_t B, int32_t C, int32_t KZ, int32_t pz, int64_t planes, int32_t Z,
                int32_t d_ctot, int32_t d_off, int32_t c_fill, int32_t dtype, void* stream);

/* BatchNorm3d (torch_blocks.py:20-25) on NDHWC tensors, fp32 statistics.
 * stats: sums[2*C] = {sum d, sum d^2}, d = x - shift[c] (shift NULL = 0).  Two calls - shift 0, then
 * shift = mean - give a cancellation-free variance.  `partials` = caller-owned scratch of
 * WSR_CHAN_SUM_ROWS * 2C floats: with it (and C a multiple of 4, <= 512) the reduction is vectorised,
 * atomic-free and OVERWRITES sums; with NULL the scalar kernel ADDS to sums (caller zeroes them).     */
int wsr_bn_stats(const void* x, int32_t C, int64_t nvox, const float* shift, float* sums, float* partials,
                 int32_t dtype, void* stream);
/* The small per-channel steps between the two statistics passes, fused (they are launch-latency bound):
 * mean = sums[0:C] / count;  then from the shifted sums: biased variance, invstd = rsqrt(var + eps) and the
 * nn.BatchNorm3d running-stat update (unbiased variance, `momentum`).  `count_dev` (device scalar, e.g. the
 * all-reduced voxel count under data parallelism) overrides `count_host` when not NULL.                  */
int wsr_bn_mean(const float* sums, const float* count_dev, float count_host, float* mean, int32_t C, void* stream);
/* ABI 5 - SyncBN around its one collective per layer (data-parallel nn.BatchNorm3d, torch_blocks.py:20-25).  G groups
 * (D(real), D(fake)) of C channels; `work` rows hold the local means at [0, C), `s2` rows the shifted sums {sum d,
 * sum d^2} at [0, 2C) (strides in floats).  _shard_stats writes this rank's record send (G, 2C) = {mean, M2};
 * _combine_shards takes the gathered (world, G, 2C) records and leaves the global mean in work[:, 0:C) and {0, M2} in
 * s2[:, 0:2C) - what wsr_bn_finalize reads with count = n * world.  Ranks are added in index order.           */
int wsr_bn_shard_stats(const float* work, int32_t work_stride, const float* s2, int32_t s2_stride, float count, float* send,
                       int32_t G, int32_t C, void* stream);
int wsr_bn_combine_shards(const float* gathered, int32_t world, float count, float* work, int32_t work_stride, float* s2,
                          int32_t s2_stride, int32_t G, int32_t C, void* stream);
int wsr_bn_finalize(const float* sums2, const float* count_dev, float count_host, const float* mean, float eps,
                    float momentum, float* invstd, float* var_out, float* running_mean, float* running_var,
                    int32_t C, void* stream);
/* y = lrelu((x-mean)*invstd*gamma + beta); mean/invstd fp32 [C]                 */
int wsr_bn_apply_lrelu(const void* x, void* y, const float* mean, const float* invstd,
                       const float* gamma, const float* beta, int32_t C, int64_t nvox, int32_t act,
                       float slope, int32_t dtype, void* stream);
/* backward, pass 1: g = dy * lrelu'(y) (written in place into dy);
 * sums[2*C] = {sum g, sum g*xhat} (`partials` as for wsr_bn_stats).              */
int wsr_bn_bwd_reduce(void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                      int32_t C, int64_t nvox, int32_t act, float slope, float* sums, float* partials,
                      int32_t dtype, void* stream);
/* pass 2 (training): dx = gamma*invstd*(g - sum_g/n - xhat*sum_gxhat/n); eval:
 * dx = gamma*invstd*g (sums == NULL).  ABI 6 - act_y != NULL: g is first multiplied by the LeakyReLU derivative
 * (act_y > 0 ? 1 : slope) of the layer's saved OUTPUT act_y (same shape) - an eval-mode layer in a pass that wants
 * no parameter gradients (D inside a generator iteration, wind_field_GAN_3D.py:570-583) needs no separate
 * leaky_relu_backward pass.                                                                                 */
int wsr_bn_bwd_apply(const void* g, const void* x, void* dx, const float* mean, const float* invstd,
                     const float* gamma, const float* sums, float inv_n, const void* act_y, float slope, int32_t C,
                     int64_t nvox, int32_t dtype, void* stream);

/* ---- optimizer ---------------------------------------------------------------
 * torch.optim.Adam single-tensor step over a flat fp32 buffer
 * (wind_field_GAN_3D.py:151-162,460,566), bias-corrected, L2 weight decay.     */
int wsr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int32_t step, void* stream);
/* ABI 7 - the same update for MANY tensors in one launch (an optimizer's whole parameter list:
 * wind_field_GAN_3D.py:460 `optimizer_G.step()`, :566 `optimizer_D.step()`): `jobs_dev` is a DEVICE array of n_jobs
 * records, one workgroup each - the caller cuts large tensors into chunks (32 768 elements is a good size).  All
 * tensors of a call share the hyper-parameters and the step count, as the parameters of one param group do.     */
typedef struct wsr_adam_job {
  float* p;       /* parameter chunk, updated in place */
  const float* g; /* its gradient                      */
  float* m;       /* exp_avg                           */
  float* v;       /* exp_avg_sq                        */
  int64_t n;      /* elements                          */
} wsr_adam_job_t;
/* (hyper-parameters as doubles, the way torch.optim.Adam holds them: 1 - beta is taken in double and rounded once, so
 *  exp_avg / exp_avg_sq follow torch's to the last bits instead of to 1e-5)                                           */
int wsr_adam_multi(const wsr_adam_job_t* jobs_dev, int32_t n_jobs, double lr, double beta1, double beta2, double eps,
                   double weight_decay, int32_t step, void* stream);

/* ABI 9 - train-mode statistics of ALL batch groups of a BatchNorm3d layer (torch_blocks.py:20-25; the groups are the
 * reference's separate calls D(real), D(fake) of one iteration, wind_field_GAN_3D.py:247-304, batched into one pass) in four
 * launches: what wsr_bn_stats + wsr_bn_mean + wsr_bn_stats(shift = mean) + wsr_bn_finalize compute per group in twelve,
 * bit for bit.  x: `groups` consecutive blocks of nvox_g voxels x C channels (NDHWC).  work: [groups][2C] floats out - mean
 * | 1 / sqrt(biased var + eps) of each group.  running_mean / running_var (NULL: untouched) are updated once per group, in
 * group order, as the groups' separate calls would (momentum; unbiased variance).  partials: caller-owned scratch of
 * groups * WSR_CHAN_SUM_ROWS * 2C floats.  C a multiple of 4, <= 512; 1 <= groups <= 64.                          */
int wsr_bn_train_stats(const void* x, int32_t C, int64_t nvox_g, int32_t groups, float eps, float momentum, float* work,
                       float* running_mean, float* running_var, float* partials, int32_t dtype, void* stream);

/* ---- relativistic average GAN loss -------------------------------------------
 * ABI 9 - the adversarial term of a generator iteration (wind_field_GAN_3D.py:360-364) and the loss of a
 * discriminator iteration (:552-556), both of the form
 *     L = ( BCEWithLogits(u - mean(v), lu) + BCEWithLogits(v - mean(u), lv) ) / 2
 * (generator: u = D(fake), v = D(real); discriminator: u = D(real), v = D(fake); lu / lv the label vectors of
 * :627-678), forward AND every partial derivative in one launch - the reference's composed ops are ~25 launches of
 * B-element kernels per loss and pass.  u, v, lu, lv: B floats each.  mu / mv: device scalars holding mean(u) /
 * mean(v) when the caller computed them (data parallelism: batch-global means from a collective), NULL: taken here
 * over the B elements and the chain rule through them folded into du / dv.  out: 2 B + 3 floats
 *     [0] L, [1 .. B] dL/du, [B+1 .. 2B] dL/dv, [2B+1] dL/dmu, [2B+2] dL/dmv   (the last two 0 when mu / mv are NULL)
 * BCEWithLogits(x, t) = mean_i (1 - t_i) x_i - logsigmoid(x_i), torch's formula.  1 <= B <= 65536.             */
int wsr_ragan_loss(const float* u, const float* v, const float* lu, const float* lv, const float* mu, const float* mv,
                   int32_t B, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WINDSR_HIP_H */
