"""Thin torch-tensor front end over the C-ABI (device pointers + current stream).

Activations are ``(B, X, Y, Z, ctot)`` contiguous tensors ("NDHWC"), fp32 or
bf16; filters are packed ``[rows][taps][k]`` tensors produced by
:func:`pack_filter`.  Every function launches on the *current* torch stream and
returns immediately; nothing here allocates behind the caller's back except
where a fresh output tensor is the documented result.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, Epilogue, check

Tensor = torch.Tensor


def dtype_id(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return _lib.WSR_F32
    if dt == torch.bfloat16:
        return _lib.WSR_BF16
    raise TypeError(f"unsupported activation dtype {dt}")


def piece_elems(dt: torch.dtype) -> int:
    """Channels per 16-byte piece (granularity of channel windows on the MFMA path)."""
    return 4 if dt == torch.float32 else 8


def pad_channels(c: int, dt: torch.dtype) -> int:
    e = piece_elems(dt)
    return (c + e - 1) // e * e


# torch.cuda.current_stream() / current_device() cost ~8 us / ~1 us of Python per call (device-index normalisation,
# lazy-init checks, a Stream object); at ~2 000 launches per step that is most of the host time of the launch-bound
# real-data shapes.  The raw accessors below are what they end in.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> C.c_void_p:
    if _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _need_cuda(*ts: Tensor) -> None:
    """Operands must live on the CURRENT device: the kernels are launched on its current stream with raw
    pointers, so a tensor of another GPU would be a wild pointer there (no CPU fallback either)."""
    cur = -1
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("windsr_hip kernels need device tensors (no CPU fallback)")
        if cur < 0:
            cur = _raw_device() if _raw_device is not None else torch.cuda.current_device()
        if t.device.index != cur:
            raise RuntimeError(f"windsr_hip: tensor on {t.device} but the current device is cuda:{cur} "
                               "(set torch.cuda.set_device / cfg.device for this rank)")


@dataclass(frozen=True)
class ConvGeom:
    """Static description of a conv layer (reference ``nn.Conv3d`` arguments)."""

    cin: int
    cout: int
    kernel: Tuple[int, int, int]
    stride: Tuple[int, int, int] = (1, 1, 1)
    pad: Tuple[int, int, int] = (1, 1, 1)
    upsample: bool = False  # nearest x(2,2,1) in front (torch_blocks.py:345-347)

    @property
    def taps(self) -> int:
        return self.kernel[0] * self.kernel[1] * self.kernel[2]

    def out_extent(self, xi: int, yi: int, zi: int) -> Tuple[int, int, int]:
        u = 2 if self.upsample else 1
        k, s, p = self.kernel, self.stride, self.pad
        return ((xi * u + 2 * p[0] - k[0]) // s[0] + 1, (yi * u + 2 * p[1] - k[1]) // s[1] + 1,
                (zi + 2 * p[2] - k[2]) // s[2] + 1)


_desc_cache: dict = {}


def make_desc(g: ConvGeom, dt: torch.dtype, B: int, in_xyz, in_ctot: int, in_off: int, out_ctot: int,
              out_off: int, cin: Optional[int] = None, cout: Optional[int] = None, lat=None) -> ConvDesc:
    """``lat`` = (ox, oy, phases): a parity conv of the sub-pixel form of an up-sampling conv (``wsr_conv_t.lat``) -
    same-size, ``g.pad`` = the LOW pads, output voxels on the (2x + ox, 2y + oy) lattice of a tensor twice as large
    along x and y; phases = 4: all four parities in one forward launch.

    The descriptor is pure geometry: it is built once per distinct argument tuple and shared (callers pass it to
    the C side by const reference and never write to it) - filling the 30 ctypes fields cost ~10 us of the ~17 us
    of Python per launch on the launch-bound real-data shapes."""
    key = (g, dt, B, tuple(in_xyz), in_ctot, in_off, out_ctot, out_off, cin, cout, None if lat is None else tuple(lat))
    d = _desc_cache.get(key)
    if d is not None:
        return d
    if len(_desc_cache) > 8192:
        _desc_cache.clear()
    d = _desc_cache[key] = _build_desc(g, dt, B, in_xyz, in_ctot, in_off, out_ctot, out_off, cin, cout, lat)
    return d


def _build_desc(g, dt, B, in_xyz, in_ctot, in_off, out_ctot, out_off, cin, cout, lat) -> ConvDesc:
    xo, yo, zo = g.out_extent(*in_xyz)
    if lat is not None:
        xo, yo, zo = in_xyz
    d = ConvDesc()
    d.dtype = dtype_id(dt)
    d.B = B
    d.Xi, d.Yi, d.Zi = in_xyz
    d.Xo, d.Yo, d.Zo = xo, yo, zo
    d.Cin, d.in_ctot, d.in_off = (g.cin if cin is None else cin), in_ctot, in_off
    d.Cout, d.out_ctot, d.out_off = (g.cout if cout is None else cout), out_ctot, out_off
    d.KX, d.KY, d.KZ = g.kernel
    d.sx, d.sy, d.sz = g.stride
    d.px, d.py, d.pz = g.pad
    d.upsample_xy = 1 if g.upsample else 0
    if lat is not None:
        d.lat, d.lat_ox, d.lat_oy, d.lat_phases = 2, lat[0], lat[1], lat[2]
        if len(lat) > 3:  # (ox, oy, phases, mz, oz): z lattice of the output as well
            d.lat_mz, d.lat_oz = lat[3], lat[4]
        if len(lat) > 5 and lat[5]:  # (..., True): the INPUT sits on the lattice (filter gradients of strided convs)
            d.lat = 3
    return d


def conv_fwd(desc: ConvDesc, x: Tensor, w: Tensor, y: Tensor, *, bias: Optional[Tensor] = None,
             chan_scale: Optional[Tensor] = None, res: Optional[Tensor] = None, res_off: int = 0,
             alpha: float = 1.0, beta: float = 0.0, act: bool = False, slope: float = 0.2,
             out_planar: bool = False) -> Tensor:
    _need_cuda(x, w, y, bias, chan_scale, res)
    ep = Epilogue()
    ep.bias, ep.chan_scale, ep.res = _p(bias), _p(chan_scale), _p(res)
    ep.res_ctot = res.shape[-1] if res is not None else 0
    ep.res_off = res_off
    ep.alpha, ep.beta = alpha, beta
    ep.act, ep.slope = int(act), slope
    ep.out_planar = int(out_planar)
    check(_lib.lib().wsr_conv3d_fwd(C.byref(desc), _p(x), _p(w), _p(y), C.byref(ep), _stream()), "conv3d_fwd")
    return y


_tile_ws: dict = {}
TILE_WS_BYTES = 32 << 20

#: bumped by :func:`reload_env`; host-side caches of values that depend on the WSR_* tuning switches (the split counts
#: of the filter-gradient plans, unpack job tables) compare it and start over
ENV_GEN = [0]


def reload_env() -> None:
    """The process changed its WSR_* environment at run time (tests, tuning scripts): have the C side read its cached
    switches again (``wsr_reload_env``) and invalidate what the host side cached from them."""
    check(_lib.lib().wsr_reload_env(), "reload_env")
    ENV_GEN[0] += 1


def tile_workspace(dev=None):
    """(pointer, bytes) of the split-reduction workspace the tile entry points get with every call (``wsr_epilogue_t.ws``,
    ``wsr_dgrad_opts_t.ws``): launches with few workgroups and a long reduction (the discriminator's deep layers) then
    spread the reduction channels over up to 256 workgroups.  One buffer per (device, stream) - launches on one
    stream are ordered, launches on different streams may overlap and must not share partial sums; the library itself
    keeps no state.  Owned here, lives as long as the process."""
    key = (_raw_device() if _raw_device is not None else torch.cuda.current_device(), _stream().value)
    ws = _tile_ws.get(key)
    if ws is None:
        ws = _tile_ws[key] = torch.empty(TILE_WS_BYTES, dtype=torch.uint8, device=f"cuda:{key[0]}")
    return ws.data_ptr(), ws.numel()


def conv_fwd_tile(desc: ConvDesc, x: Tensor, wfrag: Tensor, y: Tensor, *, bias: Optional[Tensor] = None,
                  chan_scale: Optional[Tensor] = None, res: Optional[Tensor] = None, res_off: int = 0,
                  alpha: float = 1.0, beta: float = 0.0, act=False, slope: float = 0.2,
                  out_planar: bool = False, act_c1: int = 0, res2: Optional[Tensor] = None, res2_off: int = 0,
                  beta2: float = 0.0, use_ws: bool = True, mask=None, in2: Optional[Tensor] = None,
                  in2_c0: int = 0) -> bool:
    """LDS halo-tile forward conv (bf16, stride 1).  Returns False when the shape is outside
    the tile kernels (the caller then uses :func:`conv_fwd`).  ``act`` = 2 / ``act_c1``: the two stages of a
    split dense-block conv (see ``wsr_epilogue_t``)."""
    _need_cuda(x, wfrag, y, bias, chan_scale, res)
    ep = Epilogue()
    ep.bias, ep.chan_scale, ep.res = _p(bias), _p(chan_scale), _p(res)
    ep.res_ctot = res.shape[-1] if res is not None else 0
    ep.res_off = res_off
    ep.alpha, ep.beta = alpha, beta
    ep.act, ep.slope = int(act), slope
    ep.out_planar = int(out_planar)
    ep.act_c1 = act_c1
    if res2 is not None:
        _need_cuda(res2)
        ep.res2, ep.res2_ctot, ep.res2_off, ep.beta2 = _p(res2), res2.shape[-1], res2_off, beta2
    if use_ws:
        ep.ws, ep.ws_bytes = tile_workspace()
    if in2 is not None:  # reduction channels >= in2_c0 come from this second tensor (wsr_epilogue_t.in2: the concat as two)
        _need_cuda(in2)
        if in2.shape[:-1] != x.shape[:-1] or in2.dtype != x.dtype:
            raise ValueError("in2 must have x's extents and dtype")
        ep.in2, ep.in2_ctot, ep.in2_c0 = _p(in2), in2.shape[-1], in2_c0
    if mask is not None:  # (y, y_off, c0, c1, slope): LeakyReLU-backward mask on a forward-form launch (wsr_epilogue_t.mask)
        my, my_off, mc0, mc1, mslope = mask
        _need_cuda(my)
        mk = _lib.LreluMask()
        mk.y, mk.y_ctot, mk.y_off, mk.c0, mk.c1, mk.slope = my.data_ptr(), my.shape[-1], my_off, mc0, mc1, mslope
        ep.mask = C.addressof(mk)
    rc = _lib.lib().wsr_conv3d_fwd_tile(C.byref(desc), _p(x), _p(wfrag), _p(y), C.byref(ep), _stream())
    if rc == _lib.WSR_EUNSUPPORTED:
        return False
    check(rc, "conv3d_fwd_tile")
    return True


def conv1x1_covers(red: int, n_out: int, masked: bool) -> bool:
    """shapes the streaming 1x1x1 kernel is instantiated for (conv_1x1_v2.hip, ``wsr_conv1x1_bf16``): reduction
    channels x produced channels.  Only that kernel may run a 1x1x1 input gradient in place."""
    shapes = {(128, 256)} if masked else {(256, 128), (128, 256)}
    return (red, n_out) in shapes


def conv_dgrad_tile(desc: ConvDesc, dy: Tensor, wfrag_t: Tensor, dx: Tensor, *, alpha: float = 1.0,
                    accumulate: bool = False, dx_planar: bool = False, mask=None, acc_src: Optional[Tensor] = None,
                    use_ws: bool = True, acc_beta: float = 1.0, res2: Optional[Tensor] = None, res2_off: int = 0,
                    beta2: float = 1.0, dx2: Optional[Tensor] = None, dx2_c0: int = 0) -> bool:
    """``mask`` = (y, y_off, c0, c1, slope): fold ``leaky_relu_backward`` of produced channels [c0, c1) into
    the epilogue, the mask taken from channels [y_off, y_off + c1 - c0) of the saved output ``y``.
    ``acc_src``: with ``accumulate``, the tensor (same layout as ``dx``) whose values are added instead of dx's own."""
    _need_cuda(dy, wfrag_t, dx, acc_src)
    if acc_src is not None and acc_src.shape != dx.shape:
        raise ValueError("acc_src must have dx's layout")
    opts = _lib.DgradOpts()
    opts.acc_src = _p(acc_src)
    opts.acc_beta = acc_beta
    if res2 is not None:  # (streaming 1x1x1 kernel only: the RRDB-level gradient joining the running one)
        _need_cuda(res2)
        opts.res2, opts.res2_ctot, opts.res2_off, opts.beta2 = _p(res2), res2.shape[-1], res2_off, beta2
    if use_ws:
        opts.ws, opts.ws_bytes = tile_workspace()
    if dx2 is not None:  # produced channels >= dx2_c0 go to this second tensor (wsr_dgrad_opts_t.dx2)
        _need_cuda(dx2)
        if dx2.shape[:-1] != dx.shape[:-1] or dx2.dtype != dx.dtype:
            raise ValueError("dx2 must have dx's extents and dtype")
        opts.dx2, opts.dx2_ctot, opts.dx2_c0 = _p(dx2), dx2.shape[-1], dx2_c0
    mp = None
    if mask is not None:
        y, y_off, c0, c1, slope = mask[:5]
        cs = mask[5] if len(mask) > 5 else None
        _need_cuda(y, cs)
        m = _lib.LreluMask()
        m.y, m.y_ctot, m.y_off, m.c0, m.c1, m.slope = y.data_ptr(), y.shape[-1], y_off, c0, c1, slope
        m.chan_scale = _p(cs)
        mp = C.byref(m)
    rc = _lib.lib().wsr_conv3d_dgrad_tile(C.byref(desc), _p(dy), _p(wfrag_t), _p(dx), alpha, int(accumulate),
                                          int(dx_planar), mp, C.byref(opts), _stream())
    if rc == _lib.WSR_EUNSUPPORTED:
        return False
    check(rc, "conv3d_dgrad_tile")
    return True


def pack_filter_frag(w: Tensor, *, transpose: bool = False, out: Optional[Tensor] = None,
                     dtype: torch.dtype = torch.bfloat16) -> Tensor:
    """fp32 master ``(Cout, Cin, KX, KY, KZ)`` -> MFMA-fragment order of ``dtype`` (bf16, or fp32 for the fp32 tile
    kernels) for the tile kernels."""
    _need_cuda(w)
    if not w.is_contiguous() or w.dtype != torch.float32 or w.dim() != 5:
        raise ValueError("pack_filter_frag wants a contiguous fp32 (Cout, Cin, KX, KY, KZ) tensor")
    cout, cin, kx, ky, kz = w.shape
    rows, red = (cin, cout) if transpose else (cout, cin)
    n = _lib.lib().wsr_frag_filter_elems(rows, red, kx * ky * kz, dtype_id(dtype))
    if out is None:
        out = torch.empty(n, dtype=dtype, device=w.device)
    check(_lib.lib().wsr_pack_filter_frag(_p(w), _p(out), cout, cin, kx, ky, kz, int(transpose), dtype_id(dtype),
                                          _stream()), "pack_filter_frag")
    return out


def frag_filter_elems(w: Tensor, transpose: bool, dtype: torch.dtype = torch.bfloat16) -> int:
    cout, cin, kx, ky, kz = w.shape
    rows, red = (cin, cout) if transpose else (cout, cin)
    return int(_lib.lib().wsr_frag_filter_elems(rows, red, kx * ky * kz, dtype_id(dtype)))


def frag_filter_elems_for(rows: int, red: int, taps: int, dtype: torch.dtype = torch.bfloat16) -> int:
    return int(_lib.lib().wsr_frag_filter_elems(rows, red, taps, dtype_id(dtype)))


def pack_job_table(jobs) -> Tensor:
    """Device table of ``wsr_pack_job_t`` records for ``jobs`` = [(master fp32 weight, out bf16 tensor, transpose)]
    or, for one source of a stacked dense-block filter, [(weight, out, transpose, c_lo, c_n, red_off, red_total,
    row_off, rows_total)].  The table only holds pointers and shapes, so it stays valid while those tensors keep
    their storage."""
    import numpy as np

    rec = np.zeros((len(jobs), 8), dtype=np.int64)  # 2 pointers + 12 int32
    for r, job in zip(rec, jobs):
        w, out, tr = job[:3]
        cout, cin, kx, ky, kz = w.shape
        r[0], r[1] = w.data_ptr(), out.data_ptr()
        r[2] = cout | (cin << 32)
        r[3] = kx | (ky << 32)
        r[4] = kz | (int(tr) << 32)
        if len(job) > 3:
            c_lo, c_n, red_off, red_total, row_off, rows_total = job[3:]
            bad = cout % 16 or red_total % 16 or c_lo < 0 or c_lo + c_n > cin
            if tr:
                bad = bad or red_off % 16 or red_off + cout > red_total or rows_total != c_n
            else:
                bad = bad or red_off or red_total != c_n or row_off % 16 or row_off + cout > rows_total
            if bad:
                raise ValueError("bad stacked filter part")
            r[5] = c_lo | (c_n << 32)
            r[6] = red_off | (red_total << 32)
            r[7] = row_off | (rows_total << 32)
    return _table_to_device(rec, jobs[0][0].device)


def _table_to_device(rec, dev) -> Tensor:
    """small host table -> device through pinned memory: a pageable copy would block the host until the stream has
    drained (one pipeline bubble per rebuilt table)"""
    t = torch.from_numpy(rec)
    return t.pin_memory().to(dev, non_blocking=True) if torch.device(dev).type == "cuda" else t.to(dev)


def pack_filter_frag_multi(table: Tensor, dtype: torch.dtype = torch.bfloat16) -> None:
    """One launch for all filters of a job table (see :func:`pack_job_table`); ``dtype`` of the fragment filters."""
    check(_lib.lib().wsr_pack_filter_frag_multi(_p(table), table.shape[0], dtype_id(dtype), _stream()),
          "pack_filter_frag_multi")


def conv_dgrad(desc: ConvDesc, dy: Tensor, wt: Tensor, dx: Tensor, *, alpha: float = 1.0,
               accumulate: bool = False, dx_planar: bool = False) -> Tensor:
    _need_cuda(dy, wt, dx)
    check(_lib.lib().wsr_conv3d_dgrad(C.byref(desc), _p(dy), _p(wt), _p(dx), alpha, int(accumulate),
                                      int(dx_planar), _stream()), "conv3d_dgrad")
    return dx


def conv_wgrad(desc: ConvDesc, x: Tensor, dy: Tensor, dw: Tensor) -> Tensor:
    """``dw`` (fp32 ``[Cout][taps][Cin]``) is accumulated into."""
    _need_cuda(x, dy, dw)
    if dw.dtype != torch.float32:
        raise TypeError("filter gradients are fp32")
    check(_lib.lib().wsr_conv3d_wgrad(C.byref(desc), _p(x), _p(dy), _p(dw), _stream()), "conv3d_wgrad")
    return dw


def conv_wgrad_tri(desc: ConvDesc, x: Tensor, dy: Tensor, dw: Tensor, tri_base: int, tri_step: int) -> Tensor:
    """Stacked filter gradient of a dense block's growth convs (see ``wsr_conv3d_wgrad_tri``)."""
    _need_cuda(x, dy, dw)
    if dw.dtype != torch.float32:
        raise TypeError("filter gradients are fp32")
    check(_lib.lib().wsr_conv3d_wgrad_tri(C.byref(desc), _p(x), _p(dy), _p(dw), tri_base, tri_step, _stream()),
          "conv3d_wgrad_tri")
    return dw


def conv_wgrad_nparts(desc: ConvDesc, tri_base: int = 0, tri_step: int = 0) -> int:
    """split copies the deterministic filter-gradient launch of this geometry writes (host-side query)"""
    n = C.c_int32(0)
    check(_lib.lib().wsr_conv3d_wgrad_nparts(C.byref(desc), tri_base, tri_step, C.byref(n)), "conv3d_wgrad_nparts")
    return int(n.value)


def conv_split_ok(desc: ConvDesc, c0: int) -> bool:
    """True when the forward, input-gradient and filter-gradient kernels all take this conv with its input channels
    split at ``c0`` over two tensors (``wsr_conv_split_ok``: host-side query)"""
    return bool(_lib.lib().wsr_conv_split_ok(C.byref(desc), c0))


def conv_wgrad_parts(desc: ConvDesc, x: Tensor, dy: Tensor, parts: Tensor, n_parts: int, tri_base: int = 0,
                     tri_step: int = 0, x2: Optional[Tensor] = None, x2_c0: int = 0) -> Tensor:
    """Deterministic filter gradient: ``parts`` (n_parts, Cout, taps, Cin) fp32 receives one partial sum per split
    (plain stores, nothing to zero); :func:`unpack_wgrad_reduce_multi` adds them in order.  ``x2``: the conv's input
    channels >= ``x2_c0`` live in this second tensor (``wsr_conv3d_wgrad_parts_x2``)."""
    _need_cuda(x, dy, parts)
    if parts.dtype != torch.float32 or not parts.is_contiguous() or parts.shape[0] != n_parts:
        raise ValueError("conv_wgrad_parts wants a contiguous fp32 (n_parts, Cout, taps, Cin) buffer")
    if x2 is not None:
        _need_cuda(x2)
        if tri_step or x2.shape[:-1] != x.shape[:-1] or x2.dtype != x.dtype:
            raise ValueError("x2 must have x's extents and dtype (no stacked form)")
        check(_lib.lib().wsr_conv3d_wgrad_parts_x2(C.byref(desc), _p(x), _p(x2), x2.shape[-1], x2_c0, _p(dy), _p(parts),
                                                   parts[0].numel(), n_parts, _stream()), "conv3d_wgrad_parts_x2")
        return parts
    check(_lib.lib().wsr_conv3d_wgrad_parts(C.byref(desc), _p(x), _p(dy), _p(parts), parts[0].numel(), n_parts, tri_base,
                                            tri_step, _stream()), "conv3d_wgrad_parts")
    return parts


def pack_filter(w: Tensor, dt: torch.dtype, *, transpose: bool = False, kpad: Optional[int] = None,
                out: Optional[Tensor] = None) -> Tensor:
    """fp32 master ``(Cout, Cin, KX, KY, KZ)`` -> compute copy ``[rows][taps][kpad]`` of ``dt``."""
    _need_cuda(w)
    cout, cin = w.shape[:2]
    taps = w[0, 0].numel()
    if not w.is_contiguous() or w.dtype != torch.float32:
        raise ValueError("pack_filter wants a contiguous fp32 (Cout, Cin, KX, KY, KZ) tensor")
    k = cout if transpose else cin
    kpad = pad_channels(k, dt) if kpad is None else kpad
    rows = cin if transpose else cout
    if out is None:
        out = torch.empty((rows, taps, kpad), dtype=dt, device=w.device)
    check(_lib.lib().wsr_pack_filter(_p(w), _p(out), dtype_id(dt), cout, taps, cin, int(transpose), kpad, _stream()),
          "pack_filter")
    return out


def unpack_wgrad(src: Tensor, dst: Tensor, scale: float = 1.0, accumulate: bool = True) -> None:
    """master-layout grad ``(Cout, Cin, KX, KY, KZ)`` (+)= scale * packed ``[Cout][taps][kpad]``."""
    cout, taps, kpad = src.shape
    if dst.dtype != torch.float32 or not dst.is_contiguous() or dst.shape[0] != cout:
        raise ValueError("unpack_wgrad wants a contiguous fp32 master-layout gradient")
    check(_lib.lib().wsr_unpack_wgrad(_p(src), _p(dst), cout, taps, dst.shape[1], kpad, scale, int(accumulate),
                                      _stream()), "unpack_wgrad")


def unpack_job_table(jobs) -> Tensor:
    """Device table of ``wsr_unpack_job_t`` records for ``jobs`` = [(packed src [Cout][taps][kpad] fp32,
    master-layout dst fp32, scale[, n_parts, part_stride])] - the last two for the deterministic split copies."""
    import numpy as np

    rec = np.zeros((len(jobs), 7), dtype=np.int64)  # 2 pointers + 4 int32 + float + int32 + 2 int32 + int64
    for r, job in zip(rec, jobs):
        src, dst, scale = job[:3]
        n_parts, part_stride = job[3:5] if len(job) > 3 else (0, 0)
        cout, taps, kpad = src.shape
        r[0], r[1] = src.data_ptr(), dst.data_ptr()
        r[2] = cout | (taps << 32)
        r[3] = dst.shape[1] | (kpad << 32)
        r[4] = int(np.float32(scale).view(np.int32)) & 0xFFFFFFFF  # accumulate = 0
        r[5] = n_parts
        r[6] = part_stride
    return _table_to_device(rec, jobs[0][0].device)


def unpack_wgrad_multi(table: Tensor) -> None:
    check(_lib.lib().wsr_unpack_wgrad_multi(_p(table), table.shape[0], _stream()), "unpack_wgrad_multi")


def unpack_wgrad_reduce_multi(table: Tensor) -> None:
    """jobs with split copies (``(src, dst, scale, n_parts, part_stride)`` records): ordered sum + unpack"""
    check(_lib.lib().wsr_unpack_wgrad_reduce_multi(_p(table), table.shape[0], _stream()), "unpack_wgrad_reduce_multi")


def lrelu_bwd_(g: Tensor, g_off: int, y: Tensor, y_off: int, C_: int, slope: float,
               chan_scale: Optional[Tensor] = None) -> None:
    nvox = g.numel() // g.shape[-1]
    check(_lib.lib().wsr_lrelu_bwd_inplace(_p(g), g.shape[-1], g_off, _p(y), y.shape[-1], y_off, C_, nvox, slope,
                                           _p(chan_scale), nvox // g.shape[0], dtype_id(g.dtype), _stream()),
          "lrelu_bwd")


def chan_axpby(dst: Tensor, d_off: int, src: Tensor, s_off: int, C_: int, alpha: float = 1.0,
               beta: float = 0.0) -> None:
    nvox = dst.numel() // dst.shape[-1]
    check(_lib.lib().wsr_chan_axpby(_p(dst), dst.shape[-1], d_off, _p(src), src.shape[-1], s_off, C_, nvox, alpha,
                                    beta, dtype_id(dst.dtype), _stream()), "chan_axpby")


CHAN_SUM_ROWS = 512  # WSR_CHAN_SUM_ROWS
_chan_sum_ws: dict = {}  # per-device scratch of the two-pass channel sum (stream-ordered re-use)


def chan_sum(x: Tensor, x_off: int, C_: int, out: Tensor, scale: float = 1.0) -> bool:
    """``out[c] = scale * sum_voxels x[..., x_off + c]`` (fp32, overwritten); False when the kernel does not
    cover the shape (channel counts that are not multiples of 4)."""
    _need_cuda(x, out)
    if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != C_:
        raise ValueError("chan_sum wants a contiguous fp32 output of C elements")
    nvox = x.numel() // x.shape[-1]
    ws = _chan_sum_ws.get(x.device)
    if ws is None or ws.numel() < CHAN_SUM_ROWS * C_:
        ws = _chan_sum_ws[x.device] = torch.empty(CHAN_SUM_ROWS * max(C_, 256), dtype=torch.float32, device=x.device)
    rc = _lib.lib().wsr_chan_sum(_p(x), x.shape[-1], x_off, C_, nvox, scale, _p(out), _p(ws), dtype_id(x.dtype),
                                 _stream())
    if rc == _lib.WSR_EUNSUPPORTED:
        return False
    check(rc, "chan_sum")
    return True


def chan_sum_rows(C_: int, nvox: int) -> int:
    """partial rows the first pass of :func:`chan_sum` writes for this shape (0: shape not covered)"""
    return int(_lib.lib().wsr_chan_sum_rows(C_, nvox))


def chan_sum_partials(x: Tensor, x_off: int, C_: int, partials: Tensor) -> None:
    """first pass of :func:`chan_sum` only: ``partials`` (rows, C) fp32 gets one row of sums per workgroup; the rows
    are added later (``wsr_unpack_wgrad_reduce_multi`` job with n_parts = rows)"""
    _need_cuda(x, partials)
    nvox = x.numel() // x.shape[-1]
    if partials.dtype != torch.float32 or not partials.is_contiguous() or \
            tuple(partials.shape) != (chan_sum_rows(C_, nvox), C_):
        raise ValueError("chan_sum_partials wants a contiguous fp32 (rows, C) buffer")
    check(_lib.lib().wsr_chan_sum_partials(_p(x), x.shape[-1], x_off, C_, nvox, _p(partials), dtype_id(x.dtype),
                                           _stream()), "chan_sum_partials")


#: first classifier layer of the discriminator through the streaming row kernel (WSR_LINEAR_ROWS=0: library GEMM)
LINEAR_ROWS = __import__("os").environ.get("WSR_LINEAR_ROWS", "1") != "0"


class _LinearRows(torch.autograd.Function):
    """``F.linear(x, w, b)`` for a few rows x and a very long reduction (``wsr_linear_rows``); backward = the
    matrix products autograd would issue."""

    @staticmethod
    def forward(ctx, x, w, b):
        y = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
        check(_lib.lib().wsr_linear_rows(_p(x), _p(w), _p(b), _p(y), x.shape[0], w.shape[0], x.shape[1], _stream()),
              "linear_rows")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = g @ w if ctx.needs_input_grad[0] else None
        gw = g.t() @ x if ctx.needs_input_grad[1] else None
        gb = g.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb


def linear_rows(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Optional[Tensor]:
    """``x @ w.T + b`` through ``wsr_linear_rows`` when the shape is the one it is built for (fp32, <= 8 rows, long
    contiguous reduction), else None (the caller uses ``F.linear``)."""
    if not (LINEAR_ROWS and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 2 and x.shape[0] <= 8
            and x.shape[1] % 4 == 0 and x.shape[1] >= 8192 and x.is_contiguous() and w.is_contiguous()
            and (b is None or (b.dtype == torch.float32 and b.is_contiguous()))):
        return None
    _need_cuda(x, w, b)
    return _LinearRows.apply(x, w, b)


def plane_sum(src: Tensor, out: Tensor) -> Tensor:
    """``out[c] = sum_{b, voxels} src[b, c]`` for a planar fp32 (B, C, ...) tensor (fp32, overwritten)"""
    _need_cuda(src, out)
    if src.dtype != torch.float32 or not src.is_contiguous() or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("plane_sum wants contiguous fp32 tensors")
    B, C_ = src.shape[:2]
    if out.numel() != C_:
        raise ValueError("plane_sum: one output element per channel")
    ws = _chan_sum_ws.get(src.device)
    if ws is None or ws.numel() < CHAN_SUM_ROWS * C_:
        ws = _chan_sum_ws[src.device] = torch.empty(CHAN_SUM_ROWS * max(C_, 256), dtype=torch.float32, device=src.device)
    check(_lib.lib().wsr_plane_sum(_p(src), B, C_, src[0, 0].numel(), _p(out), _p(ws), _stream()), "plane_sum")
    return out


def upsample2_bwd(dy: Tensor, dx: Tensor) -> Tensor:
    B, X, Y, Z, C_ = dx.shape
    check(_lib.lib().wsr_upsample2_bwd(_p(dy), _p(dx), B, X, Y, Z, C_, dtype_id(dx.dtype), _stream()),
          "upsample2_bwd")
    return dx


def strided_parity_filters(w: Tensor, out: Tensor, sz: int, zc: int) -> Tensor:
    """parity filters (4, Cin, Cout, 2, 2, KZp) of the input gradient of a stride-(2, 2, sz) 4x4x3 conv
    (``wsr_strided_parity_filters``): forward-conv filters over dy, rows = the conv's input channels"""
    _need_cuda(w, out)
    cout, cin = w.shape[:2]
    kzp = 3 if sz == 1 else (1 if zc == 0 else 2)
    if tuple(w.shape[2:]) != (4, 4, 3) or tuple(out.shape) != (4, cin, cout, 2, 2, kzp) or \
            not (w.is_contiguous() and out.is_contiguous()) or w.dtype != torch.float32 or out.dtype != torch.float32:
        raise ValueError("strided_parity_filters: fp32 (Cout, Cin, 4, 4, 3) -> (4, Cin, Cout, 2, 2, KZp)")
    check(_lib.lib().wsr_strided_parity_filters(_p(w), _p(out), cout, cin, sz, zc, _stream()), "strided_parity_filters")
    return out


def strided_parity_unfold(dwp: Tensor, dw: Tensor, sz: int, zc: int) -> Tensor:
    """class gradients (4, Cout, Cin, 2, 2, KZp) of the parity form of a stride-(2, 2, sz) 4x4x3 conv's filter gradient
    -> their taps of the master gradient (Cout, Cin, 4, 4, 3) (``wsr_strided_parity_unfold``)"""
    _need_cuda(dwp, dw)
    cout, cin = dw.shape[:2]
    kzp = 3 if sz == 1 else (1 if zc == 0 else 2)
    if tuple(dw.shape[2:]) != (4, 4, 3) or tuple(dwp.shape) != (4, cout, cin, 2, 2, kzp) or \
            not (dw.is_contiguous() and dwp.is_contiguous()) or dw.dtype != torch.float32 or dwp.dtype != torch.float32:
        raise ValueError("strided_parity_unfold: fp32 (4, Cout, Cin, 2, 2, KZp) -> (Cout, Cin, 4, 4, 3)")
    check(_lib.lib().wsr_strided_parity_unfold(_p(dwp), _p(dw), cout * cin, sz, zc, _stream()), "strided_parity_unfold")
    return dw


def subpixel_fold(w: Tensor, wp: Tensor) -> Tensor:
    """master fp32 (Cout, Cin, 3, 3, KZ) -> the four parity filters (4, Cout, Cin, 2, 2, KZ) of the sub-pixel form of
    nearest x(2,2,1) + conv (``wsr_subpixel_fold``)."""
    _need_cuda(w, wp)
    cout, cin, kx, ky, kz = w.shape
    if (kx, ky) != (3, 3) or tuple(wp.shape) != (4, cout, cin, 2, 2, kz) or not (w.is_contiguous() and wp.is_contiguous()) \
            or w.dtype != torch.float32 or wp.dtype != torch.float32:
        raise ValueError("subpixel_fold: fp32 (Cout, Cin, 3, 3, KZ) -> (4, Cout, Cin, 2, 2, KZ)")
    check(_lib.lib().wsr_subpixel_fold(_p(w), _p(wp), cout * cin, kz, _stream()), "subpixel_fold")
    return wp


def subpixel_unfold(dwp: Tensor, dw: Tensor) -> Tensor:
    """adjoint of :func:`subpixel_fold`: parity filter gradients (4, Cout, Cin, 2, 2, KZ) -> (Cout, Cin, 3, 3, KZ)"""
    _need_cuda(dwp, dw)
    cout, cin, kx, ky, kz = dw.shape
    if (kx, ky) != (3, 3) or tuple(dwp.shape) != (4, cout, cin, 2, 2, kz) or not (dw.is_contiguous() and dwp.is_contiguous()) \
            or dw.dtype != torch.float32 or dwp.dtype != torch.float32:
        raise ValueError("subpixel_unfold: fp32 (4, Cout, Cin, 2, 2, KZ) -> (Cout, Cin, 3, 3, KZ)")
    check(_lib.lib().wsr_subpixel_unfold(_p(dwp), _p(dw), cout * cin, kz, _stream()), "subpixel_unfold")
    return dw


class _WindGradient(torch.autograd.Function):
    """(B, 3, X, Y, Z) -> (B, 9, X, Y, Z) derivatives of the wind field (``wsr_wind_gradient``) with the adjoint
    as backward - one launch each instead of torch.gradient + slicing arithmetic and their autograd graph."""

    @staticmethod
    def forward(ctx, f: Tensor, xs: Tensor, ys: Tensor, zc: Tensor) -> Tensor:
        B, _, X, Y, Z = f.shape
        out = torch.empty((B, 9, X, Y, Z), dtype=torch.float32, device=f.device)
        check(_lib.lib().wsr_wind_gradient(_p(f), _p(xs), _p(ys), _p(zc), _p(out), B, X, Y, Z, _stream()),
              "wind_gradient")
        ctx.save_for_backward(xs, ys, zc)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        xs, ys, zc = ctx.saved_tensors
        g = g.contiguous().float()
        B, _, X, Y, Z = g.shape
        df = torch.empty((B, 3, X, Y, Z), dtype=torch.float32, device=g.device)
        check(_lib.lib().wsr_wind_gradient_bwd(_p(g), _p(xs), _p(ys), _p(zc), _p(df), B, X, Y, Z, _stream()),
              "wind_gradient_bwd")
        return df, None, None, None


def wind_gradient(f: Tensor, xs: Tensor, ys: Tensor, zc: Tensor) -> Tensor:
    """``calculate_gradient_of_wind_field`` of the reference (process_data.py:301-313) on the device"""
    _need_cuda(f, xs, ys, zc)
    B, C_, X, Y, Z = f.shape
    if C_ != 3 or xs.numel() != X or ys.numel() != Y or zc.numel() != B * X * Y * Z:
        raise ValueError("wind_gradient wants f (B,3,X,Y,Z), x (X), y (Y), Z (B,1,X,Y,Z)")
    return _WindGradient.apply(f.contiguous().float(), xs.contiguous().float(), ys.contiguous().float(),
                               zc.contiguous().float())


_physics_ws: dict = {}  # per-device partial table of the statistics pass (stream-ordered re-use)


class _PhysicsLossStats(torch.autograd.Function):
    """``wsr_physics_loss_stats`` with ``wsr_physics_loss_bwd`` as backward: (sums[6], maxima[8]) of the content
    losses in one pass over HR, SR, Z; only the sums carry gradient, and only towards SR."""

    @staticmethod
    def forward(ctx, hr: Tensor, sr: Tensor, xs: Tensor, ys: Tensor, zc: Tensor):
        B, _, X, Y, Z = sr.shape
        dev = sr.device
        ws = _physics_ws.get(dev)
        if ws is None:
            ws = _physics_ws[dev] = torch.empty(int(_lib.lib().wsr_physics_loss_workspace_floats()),
                                                dtype=torch.float32, device=dev)
        stats = torch.empty(14, dtype=torch.float32, device=dev)
        check(_lib.lib().wsr_physics_loss_stats(_p(hr), _p(sr), _p(xs), _p(ys), _p(zc), _p(stats), _p(ws), B, X, Y, Z,
                                                _stream()), "physics_loss_stats")
        ctx.save_for_backward(hr, sr, xs, ys, zc)
        sums, maxima = stats[:6], stats[6:]
        ctx.mark_non_differentiable(maxima)
        return sums, maxima

    @staticmethod
    def backward(ctx, g_sums: Tensor, _g_max):
        hr, sr, xs, ys, zc = ctx.saved_tensors
        B, _, X, Y, Z = sr.shape
        coef = g_sums.contiguous().float()
        resid = torch.empty((B, 9, X, Y, Z), dtype=torch.float32, device=sr.device)
        dsr = torch.empty_like(sr)
        check(_lib.lib().wsr_physics_loss_bwd(_p(hr), _p(sr), _p(xs), _p(ys), _p(zc), _p(coef), _p(resid), _p(dsr), B, X,
                                              Y, Z, _stream()), "physics_loss_bwd")
        return None, dsr, None, None, None


def physics_loss_stats(hr: Tensor, sr: Tensor, xs: Tensor, ys: Tensor, zc: Tensor):
    """-> (sums[6], maxima[8]), see ``wsr_physics_loss_stats`` in windsr_hip.h.  hr, sr (B,3,X,Y,Z), zc (B,1,X,Y,Z)."""
    _need_cuda(hr, sr, xs, ys, zc)
    B, C_, X, Y, Z = sr.shape
    if C_ != 3 or hr.shape != sr.shape or xs.numel() != X or ys.numel() != Y or zc.numel() != B * X * Y * Z:
        raise ValueError("physics_loss_stats wants hr, sr (B,3,X,Y,Z), x (X), y (Y), Z (B,1,X,Y,Z)")
    f = lambda t: t.contiguous().float()  # noqa: E731
    return _PhysicsLossStats.apply(f(hr), f(sr), f(xs), f(ys), f(zc))


class _RaganLoss(torch.autograd.Function):
    """( BCEWithLogits(u - mean v, lu) + BCEWithLogits(v - mean u, lv) ) / 2 with every partial derivative from ONE launch
    (``wsr_ragan_loss``); the backward is one multiply of the saved derivative vector by the upstream scalar."""

    @staticmethod
    def forward(ctx, u: Tensor, v: Tensor, lu: Tensor, lv: Tensor, mu: Optional[Tensor], mv: Optional[Tensor]):
        B = u.numel()
        flat = [t.detach().reshape(-1).float().contiguous() for t in (u, v, lu, lv)]
        if any(t.numel() != B for t in flat):
            raise ValueError("ragan_loss: logits and labels must have one element per sample")
        sc = [None if m is None else m.detach().reshape(1).float().contiguous() for m in (mu, mv)]
        _need_cuda(*flat, *sc)
        out = torch.empty(2 * B + 3, dtype=torch.float32, device=u.device)
        check(_lib.lib().wsr_ragan_loss(_p(flat[0]), _p(flat[1]), _p(flat[2]), _p(flat[3]), _p(sc[0]), _p(sc[1]), B, _p(out),
                                        _stream()), "ragan_loss")
        ctx.save_for_backward(out)
        ctx.meta = (B, u.shape, v.shape, u.dtype, v.dtype, None if mu is None else (mu.shape, mu.dtype),
                    None if mv is None else (mv.shape, mv.dtype))
        return out[0].clone()

    @staticmethod
    def backward(ctx, g: Tensor):
        (out,) = ctx.saved_tensors
        B, us, vs, ud, vd, mum, mvm = ctx.meta
        d = out[1:] * g
        du = d[:B].reshape(us).to(ud) if ctx.needs_input_grad[0] else None
        dv = d[B:2 * B].reshape(vs).to(vd) if ctx.needs_input_grad[1] else None
        dmu = d[2 * B].reshape(mum[0]).to(mum[1]) if mum is not None and ctx.needs_input_grad[4] else None
        dmv = d[2 * B + 1].reshape(mvm[0]).to(mvm[1]) if mvm is not None and ctx.needs_input_grad[5] else None
        return du, dv, None, None, dmu, dmv


def ragan_loss(u: Tensor, v: Tensor, lu: Tensor, lv: Tensor, mean_u: Optional[Tensor] = None,
               mean_v: Optional[Tensor] = None) -> Tensor:
    """Relativistic average GAN loss of the reference (wind_field_GAN_3D.py:360-364 with u = D(fake), v = D(real);
    :552-556 with u = D(real), v = D(fake)); ``mean_u`` / ``mean_v``: batch-global means from the caller's collective
    (both or neither), else the means over these samples."""
    if (mean_u is None) != (mean_v is None):
        raise ValueError("ragan_loss takes both means or neither")
    return _RaganLoss.apply(u, v, lu, lv, mean_u, mean_v)


def zfold(t: Tensor, y: Tensor, bias: Optional[Tensor], kz: int, pz: int) -> Tensor:
    """``y[b,c,x,y,z] = bias[c] + sum_k t[b, c*kz+k, x, y, z+k-pz]`` - planar fp32 (see ``wsr_zfold``)."""
    _need_cuda(t, y)
    B, C_ = y.shape[:2]
    if t.dtype != torch.float32 or y.dtype != torch.float32 or not (t.is_contiguous() and y.is_contiguous()) \
            or t.shape[1] != C_ * kz or t.shape[2:] != y.shape[2:]:
        raise ValueError("zfold wants contiguous fp32 (B, C*kz, X, Y, Z) -> (B, C, X, Y, Z)")
    Z = y.shape[-1]
    planes = y[0, 0].numel() // Z
    check(_lib.lib().wsr_zfold(_p(t), _p(y), _p(bias) if bias is not None else None, B, C_, kz, pz, planes, Z,
                               _stream()), "zfold")
    return y


def zunfold(g: Tensor, d: Tensor, kz: int, pz: int, d_off: int = 0, c_fill: Optional[int] = None) -> Tensor:
    """adjoint of :func:`zfold` into an NDHWC window: ``d[b,x,y,z, c*kz+k] = g[b,c,x,y, z-k+pz]``"""
    _need_cuda(g, d)
    if g.dtype != torch.float32 or not g.is_contiguous():
        raise ValueError("planar tensors are contiguous fp32 (B, C, X, Y, Z)")
    B, C_ = g.shape[:2]
    Z = g.shape[-1]
    planes = g[0, 0].numel() // Z
    c_fill = C_ * kz if c_fill is None else c_fill
    check(_lib.lib().wsr_zunfold(_p(g), _p(d), B, C_, kz, pz, planes, Z, d.shape[-1], d_off, c_fill,
                                 dtype_id(d.dtype), _stream()), "zunfold")
    return d


def planar_to_ndhwc(src: Tensor, dst: Tensor, d_off: int = 0, c_fill: Optional[int] = None) -> Tensor:
    """fp32 (B, C, X, Y, Z) contiguous -> channel window of an NDHWC tensor."""
    _need_cuda(src, dst)
    if src.dtype != torch.float32 or not src.is_contiguous():
        raise ValueError("planar tensors are contiguous fp32 (B, C, X, Y, Z)")
    B, C_ = src.shape[:2]
    vpb = src[0, 0].numel()
    c_fill = C_ if c_fill is None else c_fill
    check(_lib.lib().wsr_planar_to_ndhwc(_p(src), _p(dst), B, C_, vpb, dst.shape[-1], d_off, c_fill,
                                         dtype_id(dst.dtype), _stream()), "planar_to_ndhwc")
    return dst


def ndhwc_to_planar(src: Tensor, C_: int, s_off: int = 0) -> Tensor:
    B, X, Y, Z, ctot = src.shape
    dst = torch.empty((B, C_, X, Y, Z), dtype=torch.float32, device=src.device)
    check(_lib.lib().wsr_ndhwc_to_planar(_p(src), _p(dst), B, C_, X * Y * Z, ctot, s_off, dtype_id(src.dtype),
                                         _stream()), "ndhwc_to_planar")
    return dst


def _reduce_ws(dev, C_: int) -> Optional[Tensor]:
    """scratch of the atomic-free reductions (None: the scalar kernels, which accumulate, take over)"""
    if C_ % 4 or C_ > 512:
        return None
    ws = _chan_sum_ws.get(dev)
    if ws is None or ws.numel() < CHAN_SUM_ROWS * 2 * C_:
        ws = _chan_sum_ws[dev] = torch.empty(CHAN_SUM_ROWS * max(2 * C_, 256), dtype=torch.float32, device=dev)
    return ws


def bn_stats(x: Tensor, sums: Tensor, shift: Optional[Tensor] = None) -> None:
    """sums[2C] = {sum (x - shift), sum (x - shift)^2} per channel"""
    C_ = x.shape[-1]
    ws = _reduce_ws(x.device, C_)
    if ws is None:
        sums.zero_()
    check(_lib.lib().wsr_bn_stats(_p(x), C_, x.numel() // C_, _p(shift), _p(sums), _p(ws), dtype_id(x.dtype),
                                  _stream()), "bn_stats")


_bn_group_ws: dict = {}


def bn_train_stats(x: Tensor, groups: int, work: Tensor, eps: float, momentum: float, running_mean: Optional[Tensor],
                   running_var: Optional[Tensor]) -> bool:
    """Train-mode BatchNorm statistics of ``groups`` equal batch groups of ``x`` (NDHWC) in four launches
    (``wsr_bn_train_stats``): ``work[g] = (mean | invstd)`` of group g; the running statistics are updated once per group,
    in order.  False: shape outside the vectorised kernels (the caller runs the per-group ops)."""
    _need_cuda(x, work, running_mean, running_var)
    C_ = x.shape[-1]
    if C_ % 4 or C_ > 512 or x.shape[0] % groups or not x.is_contiguous() or groups > 64:
        return False
    if work.dtype != torch.float32 or tuple(work.shape) != (groups, 2 * C_) or not work.is_contiguous():
        raise ValueError("bn_train_stats wants a contiguous fp32 (groups, 2C) work buffer")
    nvox_g = x.numel() // C_ // groups
    key = (x.device, groups)
    ws = _bn_group_ws.get(key)
    if ws is None or ws.numel() < groups * CHAN_SUM_ROWS * 2 * C_:
        ws = _bn_group_ws[key] = torch.empty(groups * CHAN_SUM_ROWS * 2 * max(C_, 128), dtype=torch.float32, device=x.device)
    check(_lib.lib().wsr_bn_train_stats(_p(x), C_, nvox_g, groups, eps, momentum, _p(work), _p(running_mean), _p(running_var),
                                        _p(ws), dtype_id(x.dtype), _stream()), "bn_train_stats")
    return True


def bn_mean(sums: Tensor, mean: Tensor, count: float, count_dev: Optional[Tensor] = None) -> None:
    check(_lib.lib().wsr_bn_mean(_p(sums), _p(count_dev), float(count), _p(mean), mean.numel(), _stream()), "bn_mean")


def bn_shard_stats(work: Tensor, s2: Tensor, count: float, send: Tensor) -> None:
    """SyncBN: this rank's record ``send`` (G, 2C) = {local mean, local centred second moment} from ``work`` (G, >= C:
    the means) and ``s2`` (G, >= 2C: the shifted sums) - rows may be strided views."""
    G_, C_ = send.shape[0], send.shape[1] // 2
    if work.stride(-1) != 1 or s2.stride(-1) != 1 or not send.is_contiguous():
        raise ValueError("bn_shard_stats wants unit-stride rows")
    check(_lib.lib().wsr_bn_shard_stats(_p(work), work.stride(0), _p(s2), s2.stride(0), float(count), _p(send), G_, C_,
                                        _stream()), "bn_shard_stats")


def bn_combine_shards(gathered: Tensor, count: float, work: Tensor, s2: Tensor) -> None:
    """SyncBN: ``gathered`` (world, G, 2C) records of all ranks -> global mean in ``work[:, :C]``, {0, M2} in
    ``s2[:, :2C]`` (what :func:`bn_finalize` reads with ``count * world``)."""
    world, G_, C2 = gathered.shape
    if work.stride(-1) != 1 or s2.stride(-1) != 1 or not gathered.is_contiguous():
        raise ValueError("bn_combine_shards wants unit-stride rows")
    check(_lib.lib().wsr_bn_combine_shards(_p(gathered), world, float(count), _p(work), work.stride(0), _p(s2),
                                           s2.stride(0), G_, C2 // 2, _stream()), "bn_combine_shards")


def bn_finalize(sums2: Tensor, mean: Tensor, invstd: Tensor, count: float, eps: float, momentum: float,
                running_mean: Optional[Tensor], running_var: Optional[Tensor],
                count_dev: Optional[Tensor] = None) -> None:
    check(_lib.lib().wsr_bn_finalize(_p(sums2), _p(count_dev), float(count), _p(mean), eps, momentum, _p(invstd),
                                     C.c_void_p(0), _p(running_mean), _p(running_var), mean.numel(), _stream()),
          "bn_finalize")


def bn_apply_lrelu(x: Tensor, y: Tensor, mean: Tensor, invstd: Tensor, gamma: Tensor, beta: Tensor, act: bool,
                   slope: float) -> None:
    C_ = x.shape[-1]
    check(_lib.lib().wsr_bn_apply_lrelu(_p(x), _p(y), _p(mean), _p(invstd), _p(gamma), _p(beta), C_,
                                        x.numel() // C_, int(act), slope, dtype_id(x.dtype), _stream()), "bn_apply")


def bn_bwd_reduce(dy: Tensor, y: Tensor, x: Tensor, mean: Tensor, invstd: Tensor, act: bool, slope: float,
                  sums: Tensor) -> None:
    C_ = x.shape[-1]
    ws = _reduce_ws(x.device, C_)
    if ws is None:
        sums.zero_()
    check(_lib.lib().wsr_bn_bwd_reduce(_p(dy), _p(y), _p(x), _p(mean), _p(invstd), C_, x.numel() // C_, int(act),
                                       slope, _p(sums), _p(ws), dtype_id(x.dtype), _stream()), "bn_bwd_reduce")


def bn_bwd_apply(g: Tensor, x: Tensor, dx: Tensor, mean: Tensor, invstd: Tensor, gamma: Tensor,
                 sums: Optional[Tensor], inv_n: float, act_y: Optional[Tensor] = None, slope: float = 0.2) -> bool:
    """``act_y``: fold the LeakyReLU derivative of the layer's saved output into the pass (bf16, C % 8 == 0);
    returns False when that form is not available (nothing was launched: run ``lrelu_bwd_`` and call again without)"""
    C_ = x.shape[-1]
    rc = _lib.lib().wsr_bn_bwd_apply(_p(g), _p(x), _p(dx), _p(mean), _p(invstd), _p(gamma), _p(sums), inv_n, _p(act_y),
                                     slope, C_, x.numel() // C_, dtype_id(x.dtype), _stream())
    if rc == _lib.WSR_EUNSUPPORTED and act_y is not None:
        return False
    check(rc, "bn_bwd_apply")
    return True


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, lr: float, beta1: float, beta2: float, eps: float,
              weight_decay: float, step: int) -> None:
    for t in (p, g, m, v):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("adam_step wants flat contiguous fp32 buffers")
    check(_lib.lib().wsr_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                                   _stream()), "adam_step")


ADAM_CHUNK = 32768  # elements per job of wsr_adam_multi (one workgroup each)


def adam_job_table(tensors) -> Tensor:
    """Device table of ``wsr_adam_job_t`` records for ``tensors`` = [(param, grad, exp_avg, exp_avg_sq)] (contiguous fp32,
    one device), large tensors cut into chunks of :data:`ADAM_CHUNK` elements."""
    import numpy as np

    rows = []
    for quad in tensors:
        n = quad[0].numel()
        for t in quad:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n or t.device != quad[0].device:
                raise ValueError("adam_job_table wants contiguous fp32 tensors of one size on one device")
        ptrs = [t.data_ptr() for t in quad]
        for off in range(0, n, ADAM_CHUNK):
            rows.append([q + 4 * off for q in ptrs] + [min(ADAM_CHUNK, n - off)])
    rec = np.asarray(rows, dtype=np.int64).reshape(-1, 5)
    return _table_to_device(rec, tensors[0][0].device)


def adam_multi(table: Tensor, lr: float, beta1: float, beta2: float, eps: float, weight_decay: float, step: int) -> None:
    """torch.optim.Adam's update of every tensor in ``table`` (:func:`adam_job_table`) in ONE launch; ``step`` >= 1 is the
    step count after this update (shared by all tensors, as in one param group)."""
    check(_lib.lib().wsr_adam_multi(_p(table), table.shape[0], lr, beta1, beta2, eps, weight_decay, step, _stream()),
          "adam_multi")
