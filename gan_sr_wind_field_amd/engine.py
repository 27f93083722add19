"""Fused HIP programs for Generator_3D and Discriminator_3D.features.

A *program* walks the parameter containers of a network once per call and issues
C-ABI kernel launches on the current HIP stream:

* activations live in NDHWC buffers of the compute dtype (fp32 or bf16);
* a residual dense block is one ``nf + 4*gc``-channel buffer whose channel windows
  the convs read and append to (no ``torch.cat``, reference torch_blocks.py:212-214);
* LeakyReLU, bias, ``*scale + x`` residuals, the Dropout3d channel mask, the
  nearest x(2,2,1) up-sampling and the final planar fp32 store are conv
  prologues / epilogues;
* backward is hand-derived (dgrad / wgrad kernels, LeakyReLU masks recomputed
  from the saved outputs) and returns all parameter gradients as views of ONE flat
  fp32 buffer, which is also the data-parallel all-reduce bucket space.

The programs are wrapped in ``torch.autograd.Function`` so the reference-style
train step (``loss.backward()``, ``torch.optim.Adam``) drives them unchanged.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import hip_ops as ops
from .hip_ops import ConvGeom

Tensor = torch.Tensor

#: debugging aid (tests set it): allocate work buffers filled with NaN instead of uninitialised
#: group the input gradients of a dense block by produced window (WSR_STACK_DGRAD=0: one launch per conv)
STACK_DGRAD = __import__("os").environ.get("WSR_STACK_DGRAD", "1") != "0"
#: split the growth convs of a dense block into one conv over the block input + narrow convs over the growth
#: channels (WSR_STACK_FWD=0: one launch per conv over its whole input window)
STACK_FWD = __import__("os").environ.get("WSR_STACK_FWD", "1") != "0"
#: ... and the growth-channel part of that forward grouped by SOURCE window (round 6, measured and NOT kept: WSR_FWD_REGROUP=1 turns
#: it on): the window conv j - 1 just produced is contracted into the windows of all later convs at once - 96 / 64 / 32 outputs at a
#: reduction of one window - instead of every conv re-reading every earlier window (32 outputs at 1 / 2 / 3 windows).  Same FLOPs, fewer
#: LDS fragment reads and halo bytes per MFMA; at C3' the three launches take 37.2 + 27.5 + 16.6 = 81.3 us against 72 us for the per-conv
#: stages (step 88.8 vs 87.6 ms, profiles/r06_g_ab_fwd_regroup.txt): prologue and epilogue grow with the produced width (a 96-wide
#: epilogue re-reads and re-writes three windows of partial sums) and eat what the wider K-steps save.
FWD_REGROUP = __import__("os").environ.get("WSR_FWD_REGROUP", "0") == "1"
#: train-mode BatchNorm statistics of all batch groups of a discriminator layer in four launches (wsr_bn_train_stats, round 6;
#: WSR_FUSED_BN_STATS=0: six launches per group); single-process runs only - SyncBN has its collective between the passes
FUSED_BN_STATS = __import__("os").environ.get("WSR_FUSED_BN_STATS", "1") != "0"
#: keep the running gradient of a dense block's output in channels [0, nf) of the dense gradient buffer
#: (WSR_GD_INPLACE=0: separate tensor + one add per block)
GD_INPLACE = __import__("os").environ.get("WSR_GD_INPLACE", "1") != "0"
#: the gradient of an RRDB's output stays in one dense gradient buffer while its chain runs in a second one and joins
#: the chain's result as a second residual of the last LFF input gradient: no copy / scale / add passes at the RRDB
#: boundaries (WSR_GD_PINGPONG=0: one buffer + two channel-window passes per RRDB)
GD_PINGPONG = __import__("os").environ.get("WSR_GD_PINGPONG", "1") != "0"
#: run the last conv of the generator in its z-folded form (WSR_ZFOLD=0: plain 5x5x5 conv with 3 outputs)
ZFOLD = __import__("os").environ.get("WSR_ZFOLD", "1") != "0"
#: filter gradients of the generator's dense blocks on a second HIP stream (they depend only on saved activations and the
#: block's output gradients and feed nothing in the input-gradient chain): 0 = in line on the compute stream, N >= 2 =
#: side stream with a ring of N dense gradient buffers (the running block gradient moves from buffer to buffer so that
#: a filter-gradient launch may still read the previous one)
WGRAD_STREAM = int(__import__("os").environ.get("WSR_WGRAD_STREAM", "0"))
#: fp32 programs (the reference's own arithmetic) on the LDS halo-tile kernels too: stride-1 convs, dense-block stacking,
#: the z-folded last conv (WSR_F32_TILE=0: generic implicit-GEMM kernels as in rounds 1-3)
F32_TILE = __import__("os").environ.get("WSR_F32_TILE", "1") != "0"
#: the LeakyReLU backward of the discriminator's first conv in the epilogue of the strided input gradient above it
#: (WSR_FOLD_D_MASK=1; default off: measured equal - same-device A/B 97.5 / 97.5 against 97.5 / 98.0 ms per step - the
#: masked 32-wide parity launches grow by what the pass over the 128^3 x 32 tensor costs; parity-tested either way)
FOLD_D_MASK = __import__("os").environ.get("WSR_FOLD_D_MASK", "0") != "0"
# Filter gradient of the z-folded last conv with the operands' roles exchanged (GeneratorProgram.backward): the 16-channel
# output gradient is the image that is shifted per tap, the 144-channel activation the one that is read once
SWAP_THIN_WGRAD = __import__("os").environ.get("WSR_THIN_WGRAD_SWAP", "1") != "0"
#: the generator's concat in front of the 5x5x5 conv (reference Generator_3D_Resnet_ESRGAN.py:228: torch.cat((x, Zf), 1)) as
#: TWO dense tensors - the last up-conv's nf channels and the terrain branch's tf - that the conv's three kernels read /
#: write side by side (wsr_epilogue_t.in2, wsr_dgrad_opts_t.dx2, wsr_conv3d_wgrad_parts_x2) where they can
#: (wsr_conv_split_ok), instead of one (nf + tf)-channel buffer into whose 288-byte voxel rows the terrain conv wrote
#: 32-byte pieces: partial cache lines, 1.6 x the bytes at the memory side, 0.17-0.27 of the HBM rate on the two
#: memory-bound launches that touch that window (WSR_SPLIT_CAT=0: the channel window)
SPLIT_CAT = __import__("os").environ.get("WSR_SPLIT_CAT", "1") != "0"
POISON_BUFFERS = bool(int(__import__("os").environ.get("WSR_POISON_BUFFERS", "0")))
#: filter gradients without float atomics: every spatial split of a wgrad launch stores its partial sums to its own
#: copy and the unpack pass adds the copies in index order - two backward passes give bit-identical gradients
#: (WSR_DETERMINISTIC=0: one shared copy, float atomics)
DETERMINISTIC = __import__("os").environ.get("WSR_DETERMINISTIC", "1") != "0"
#: ... their split copies live in a persistent per-program arena of this many MB that is recycled whenever it is
#: full: the pending copies are reduced into the master gradients (one launch) and the arena starts over.  (Round 2
#: sized it as the sum of all copies of a backward pass: 4.2 GB at the benchmark shape, never released.)
ARENA_MB = int(__import__("os").environ.get("WSR_ARENA_MB", "1024"))
#: input gradients of the discriminator's stride-(2,2,s) 4x4x3 convs as parity convs over dy on the tile kernels
#: (WSR_STRIDED_DGRAD=0: generic implicit-GEMM kernel)
STRIDED_DGRAD = __import__("os").environ.get("WSR_STRIDED_DGRAD", "1") != "0"
#: ... and their filter gradients: four (eight) stride-1 2x2xKZ' gradients over parity sub-lattices of the input on the
#: tile kernel (WSR_STRIDED_WGRAD=0: generic per-tap kernel, which re-reads x and dy once per tap)
STRIDED_WGRAD = __import__("os").environ.get("WSR_STRIDED_WGRAD", "1") != "0"
#: run the up-sampling convs (nearest x(2,2,1) + 3x3x3) in their sub-pixel form: four 2x2x3 parity convs on the
#: un-sampled input, 4/9 of the multiply-adds (WSR_SUBPIXEL=0: gather through the up-sampling, 27 taps)
SUBPIXEL = __import__("os").environ.get("WSR_SUBPIXEL", "1") != "0"
#: ... and their filter gradients too (WSR_SUBPIXEL_WGRAD=0: the 27-tap gradient through the up-sampling gather)
SUBPIXEL_WGRAD = __import__("os").environ.get("WSR_SUBPIXEL_WGRAD", "1") != "0"


def compute_dtype_of(flag) -> torch.dtype:
    """Map the reference's ``use_mixed_precision`` ctor flag / a dtype / a string."""
    if isinstance(flag, torch.dtype):
        return flag
    if isinstance(flag, str):
        return {"fp32": torch.float32, "f32": torch.float32, "bf16": torch.bfloat16}[flag.lower()]
    return torch.bfloat16 if flag else torch.float32


class FilterCache:
    """Packed (and transposed) compute copies of the fp32 master filters, re-packed
    only when a parameter changed (optimizer step / load_state_dict)."""

    def __init__(self, frag_dt: torch.dtype = torch.bfloat16):
        self._c: Dict[tuple, tuple] = {}
        self._tables: Dict[tuple, tuple] = {}
        self._gen = 0  # bumped by invalidate(); part of every stamp
        self.frag_dt = frag_dt  # element type of the fragment-order copies (the program's compute dtype)

    def get(self, p: Tensor, dt: torch.dtype, transpose: bool, kpad: int, rows_pad: int) -> Tensor:
        key = (id(p), dt, transpose, kpad, rows_pad)
        stamp = (p._version, p.data_ptr(), p.device, self._gen)
        hit = self._c.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        w = p.detach()
        if w.dim() == 2:  # never used for Linear; guard
            raise ValueError("filters are 5-D")
        w = w.contiguous()
        rows = w.shape[1] if transpose else w.shape[0]
        taps = w[0, 0].numel()
        if rows_pad > rows:
            out = torch.zeros((rows_pad, taps, kpad), dtype=dt, device=w.device)
        else:
            out = torch.empty((rows, taps, kpad), dtype=dt, device=w.device)
        ops.pack_filter(w, dt, transpose=transpose, kpad=kpad, out=out)
        self._c[key] = (stamp, out)
        return out

    def get_frag(self, p: Tensor, transpose: bool) -> Tensor:
        """bf16 MFMA-fragment-order copy for the LDS-tile kernels"""
        key = (id(p), "frag", transpose)
        stamp = (p._version, p.data_ptr(), p.device, self._gen)
        hit = self._c.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        out = ops.pack_filter_frag(p.detach().contiguous(), transpose=transpose, dtype=self.frag_dt)
        self._c[key] = (stamp, out)
        return out

    def refresh_frags(self, wanted, stacked=()) -> None:
        """Re-pack, in ONE launch, the fragment-order copies among ``wanted`` = [(param, transpose)] when any
        of them is stale (after an optimizer step that is all ~600 filters of the generator).  The device job
        table is cached: it only holds pointers, which stay put while parameters and copies keep their storage.
        ``stacked`` = [(key, parts, rows, red_total)] adds the stacked filters of dense blocks (``parts`` =
        [(param, transpose, c_lo, c_n, red_off, row_off)], see ``wsr_pack_job_t``); fetch them with
        :meth:`get_stacked`."""
        # cheap staleness probe first: one pass over the versions (an optimizer step bumps all of them)
        kind = ("pack", len(wanted), sum(1 for _, tr in wanted if tr), len(stacked))
        probe = (sum(p._version for p, _ in wanted), wanted[0][0].data_ptr(), wanted[-1][0].data_ptr())
        if self._tables.get(("probe",) + kind) == probe:
            return
        self._tables[("probe",) + kind] = probe
        ptrs = [p.data_ptr() for p, _ in wanted] + [p.data_ptr() for _, parts, _, _ in stacked for p, *_ in parts]
        key = tuple(ptrs) + tuple(tr for _, tr in wanted)
        cached = self._tables.get(kind)
        if cached is None or cached[0] != key:
            jobs = []
            for p, tr in wanted:
                w = p.detach()
                if not w.is_contiguous():
                    raise ValueError("conv filters must be contiguous")
                hit = self._c.get((id(p), "frag", tr))
                n = ops.frag_filter_elems(w, tr, self.frag_dt)
                out = hit[1] if hit is not None and hit[1].numel() == n and hit[1].device == w.device else \
                    torch.empty(n, dtype=self.frag_dt, device=w.device)
                jobs.append((w, out, tr))
            outs = [j[1] for j in jobs]
            for skey, parts, rows, red_total in stacked:
                w0 = parts[0][0].detach()
                n = ops.frag_filter_elems_for(rows, red_total, w0[0, 0].numel(), self.frag_dt)
                hit = self._c.get((skey, "dstack"))
                out = hit[1] if hit is not None and hit[1].numel() == n and hit[1].device == w0.device else \
                    torch.empty(n, dtype=self.frag_dt, device=w0.device)
                for p, tr, c_lo, c_n, red_off, row_off in parts:
                    w = p.detach()
                    if not w.is_contiguous():
                        raise ValueError("conv filters must be contiguous")
                    jobs.append((w, out, tr, c_lo, c_n, red_off, red_total, row_off, rows))
                outs.append(out)
            cached = (key, ops.pack_job_table(jobs), outs)
            self._tables[kind] = cached
        ops.pack_filter_frag_multi(cached[1], self.frag_dt)
        for (p, tr), out in zip(wanted, cached[2]):
            self._c[(id(p), "frag", tr)] = ((p._version, p.data_ptr(), p.device, self._gen), out)
        for (skey, _, _, _), out in zip(stacked, cached[2][len(wanted):]):
            self._c[(skey, "dstack")] = (None, out)

    def get_stacked(self, skey) -> Tensor:
        return self._c[(skey, "dstack")][1]

    def drop_tables(self) -> None:
        """forget the cached pack job tables (their destination pointers): a destination buffer was replaced"""
        self._tables.clear()

    def touch(self) -> None:
        """a source of the packed copies was rewritten behind torch's back (kernel writing through a raw pointer:
        no version bump): the next refresh_frags re-packs"""
        for k in [k for k in self._tables if k[0] == "probe"]:
            del self._tables[k]

    def invalidate(self) -> None:
        """Mark every compute copy stale (called from an optimizer post-step hook: the fused multi-tensor
        Adam updates parameters without touching their version counters)."""
        self._gen += 1
        for k in [k for k in self._tables if k[0] == "probe"]:
            del self._tables[k]

    def clear(self):
        self._c.clear()
        self._tables.clear()


@dataclass
class ConvSite:
    """One convolution of a program: parameter + static geometry."""

    name: str                 # state_dict key prefix (without .weight)
    weight: nn.Parameter
    bias: Optional[nn.Parameter]
    kernel: Tuple[int, int, int]
    stride: Tuple[int, int, int] = (1, 1, 1)
    pad: Tuple[int, int, int] = (1, 1, 1)
    upsample: bool = False
    fwd_only: bool = False    # no transposed compute copy is ever needed (parity filters of a strided input gradient)

    @property
    def cin(self) -> int:
        return self.weight.shape[1]

    @property
    def cout(self) -> int:
        return self.weight.shape[0]

    @property
    def taps(self) -> int:
        return self.kernel[0] * self.kernel[1] * self.kernel[2]


def site_from_conv(name: str, conv: nn.Conv3d, upsample: bool = False) -> ConvSite:
    return ConvSite(name, conv.weight, conv.bias, tuple(conv.kernel_size), tuple(conv.stride),
                    tuple(conv.padding), upsample)


class GradSpace:
    """Flat fp32 gradient buffer of a program; parameter gradients are views.

    Slots are laid out in *backward production order* so that contiguous ranges
    become final early and can be all-reduced while the rest of backward runs.
    """

    def __init__(self, params: Sequence[nn.Parameter]):
        self.params = list(params)
        self.offsets: Dict[int, Tuple[int, int]] = {}
        off = 0
        for p in self.params:
            n = p.numel()
            self.offsets[id(p)] = (off, n)
            off += (n + 63) // 64 * 64  # 256-byte aligned slots
        self.total = off

    def new(self, device) -> Tensor:
        return torch.empty(self.total, dtype=torch.float32, device=device)

    def view(self, flat: Tensor, p: nn.Parameter) -> Tensor:
        off, n = self.offsets[id(p)]
        return flat[off:off + n].view(p.shape)


class ProgramBase:
    def __init__(self, dt: torch.dtype):
        self.dt = dt
        self.e = ops.piece_elems(dt)
        self.filters = FilterCache(dt)
        #: optional hook(flat_grad, lo, hi, flush) called when grad range [lo, hi) is final once flush() has run
        self.grad_ready_hook: Optional[Callable[[Tensor, int, int, Callable[[], None]], None]] = None
        #: optional hook() called at the end of backward (flush + wait for gradient collectives)
        self.grad_done_hook: Optional[Callable[[], None]] = None
        #: optional hook(tag, fn) used by bench.py to time selected launches; fn() issues them
        self.launch_probe: Optional[Callable[[str, Callable[[], None]], None]] = None
        #: route stride-1 bf16 convs through the LDS halo-tile kernels (False: generic implicit GEMM only)
        self.use_tile = True
        #: test aid: a list that receives (tag, layer index, copy of the tensor) for every intermediate gradient of a
        #: backward pass - the layer-by-layer checks of the bf16 path feed each stage's fp32 CPU evaluation with the HIP
        #: path's OWN operands, so that one stage's rounding (or LeakyReLU branch flips) cannot hide another's error
        self.trace: Optional[list] = None
        self._arena: Optional[Tensor] = None
        self._arena_off = 0
        self._arena_need = 0
        self._arena_dev = None
        self._pending_unpack: list = []
        self._unpack_tables: Dict[tuple, Tensor] = {}
        self._scratch_elems_total = 0
        self._stack_specs = None
        self._stack_fwd_specs = None
        self._nparts: Dict[tuple, int] = {}
        self._env_gen = ops.ENV_GEN[0]
        self._side: Optional[torch.cuda.Stream] = None
        self._events: List[torch.cuda.Event] = []
        self._events_used = 0

    def side_stream(self, dev) -> "torch.cuda.Stream":
        """the program's second HIP stream (made once per device)"""
        if self._side is None or self._side.device != torch.device(dev):
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    def _event(self) -> "torch.cuda.Event":
        """an event from the program's pool (re-used every pass: the sequence of fences is the same)"""
        if self._events_used == len(self._events):
            self._events.append(torch.cuda.Event())
        ev = self._events[self._events_used]
        self._events_used += 1
        return ev

    def cp(self, c: int) -> int:
        """channel count padded to whole 16-byte pieces"""
        return (c + self.e - 1) // self.e * self.e

    # ---- conv helpers --------------------------------------------------------
    def tile_dt(self) -> bool:
        """the compute dtype has LDS halo-tile kernels: bf16, and fp32 (stride-1 convs; WSR_F32_TILE=0 turns it off)"""
        return self.use_tile and (self.dt == torch.bfloat16 or (self.dt == torch.float32 and F32_TILE))

    def refresh_filters(self, backward: bool) -> None:
        """bring the fragment-order filter copies of all tile-kernel convs up to date in one launch"""
        if not self.tile_dt():
            return
        sites = [s for s in self.conv_sites() if self.tile_ok(s)]
        fstack = list(self.stacked_fwd_specs())
        fcov = {id(p) for _, parts, _, _ in fstack for p, *_ in parts}
        wanted = [(s.weight, False) for s in self.conv_sites() if self.tile_fwd_ok(s) and id(s.weight) not in fcov
                  and (backward or not s.fwd_only)]  # (fwd_only: filters of a backward pass in forward-conv form)
        stacked = fstack
        if backward:
            dstack = list(self.stacked_dgrad_specs())
            covered = {id(p) for _, parts, _, _ in dstack for p, *_ in parts}
            wanted += [(s.weight, True) for s in sites if id(s.weight) not in covered and not s.fwd_only]
            stacked = fstack + dstack
        self.filters.refresh_frags(wanted, stacked)

    def stacked_dgrad_specs(self):
        return ()

    def stacked_fwd_specs(self):
        return ()

    def conv_sites(self) -> Sequence[ConvSite]:
        return getattr(self, "all_sites", [])

    def tile_ok(self, s: ConvSite) -> bool:
        """LDS halo-tile kernels: stride 1 (everything in G; the k3 s1 convs of D); fp32: no 1x1x1 (generic kernel)"""
        return self.tile_dt() and s.stride == (1, 1, 1) and (self.dt == torch.bfloat16 or s.taps > 1)

    def tile_fwd_ok(self, s: ConvSite) -> bool:
        """forward only: also the stride-2 down-sampling convs of D"""
        return self.tile_ok(s) or (self.use_tile and self.dt == torch.bfloat16 and max(s.stride) <= 2
                                   and not s.upsample)

    def _seed_parity_frags(self, par: Sequence[ConvSite], dev) -> bool:
        """The forward fragment filters of the four parity convs of one launch must sit in ONE buffer, parity-major
        (``wsr_conv_t.lat_phases``): seed the cache with views of it, refresh_frags re-packs into them.  True when a
        new buffer was made."""
        n = ops.frag_filter_elems(par[0].weight, False, self.dt)
        hits = [self.filters._c.get((id(s.weight), "frag", False)) for s in par]
        if all(h is not None and h[1].device == torch.device(dev) and h[1].numel() == n and
               h[1].data_ptr() == hits[0][1].data_ptr() + h[1].element_size() * n * ph for ph, h in enumerate(hits)):
            return False
        big = torch.empty(len(par) * n, dtype=self.dt, device=dev)
        for ph, s in enumerate(par):
            self.filters._c[(id(s.weight), "frag", False)] = (None, big[ph * n:(ph + 1) * n])
        self.filters.drop_tables()  # (cached job tables hold the old destinations)
        return True

    def _parity_frags(self, par: Sequence[ConvSite]):
        """(fragment filters of the parity sites, whether they are contiguous parity-major)"""
        frs = [self.filters.get_frag(s.weight, False) for s in par]
        n = frs[0].numel()
        return frs, all(f.data_ptr() == frs[0].data_ptr() + f.element_size() * n * ph for ph, f in enumerate(frs))

    def _w(self, s: ConvSite) -> Tensor:
        return self.filters.get(s.weight, self.dt, False, self.cp(s.cin), s.cout)

    def _wt(self, s: ConvSite) -> Tensor:
        return self.filters.get(s.weight, self.dt, True, self.cp(s.cout), self.cp(s.cin))

    def _desc(self, s: ConvSite, B: int, in_xyz, in_ctot: int, in_off: int, out_ctot: int, out_off: int,
              cin: Optional[int] = None, cout: Optional[int] = None):
        g = ConvGeom(s.cin, s.cout, s.kernel, s.stride, s.pad, s.upsample)
        return ops.make_desc(g, self.dt, B, in_xyz, in_ctot, in_off, out_ctot, out_off, cin=cin, cout=cout)

    def conv(self, s: ConvSite, x: Tensor, x_off: int, y: Tensor, y_off: int, **ep) -> None:
        """forward conv reading window [x_off, x_off+cp(cin)) of x, writing [y_off, y_off+cout) of y"""
        B = x.shape[0]
        planar = ep.get("out_planar", False)
        d = self._desc(s, B, tuple(x.shape[1:4]), x.shape[-1], x_off, s.cout if planar else y.shape[-1],
                       0 if planar else y_off, cin=self.cp(s.cin))
        bias = s.bias.detach() if s.bias is not None else None

        def run():
            if self.tile_fwd_ok(s) and ops.conv_fwd_tile(d, x, self.filters.get_frag(s.weight, False), y, bias=bias,
                                                         **ep):
                return
            if "res2" in ep:
                raise RuntimeError("a second residual needs the streaming 1x1x1 kernel")
            if ep.get("in2") is not None:
                raise RuntimeError("a two-tensor input needs the tile kernels wsr_conv_split_ok vouched for (set WSR_SPLIT_CAT=0)")
            ops.conv_fwd(d, x, self._w(s), y, bias=bias, **ep)

        if self.launch_probe is not None:
            self.launch_probe("fwd:" + s.name, run)
        else:
            run()

    def dgrad(self, s: ConvSite, g: Tensor, g_off: int, dx: Tensor, dx_off: int, in_xyz, *, alpha: float = 1.0,
              accumulate: bool = False, dx_planar: bool = False, mask=None, acc_src: Optional[Tensor] = None,
              acc_beta: float = 1.0, res2: Optional[Tensor] = None, dx2: Optional[Tensor] = None, dx2_c0: int = 0) -> None:
        """dx[window] (+)= alpha * conv^T(g[window]);  in_xyz = stored input extent of the conv.
        ``mask`` = (y, y_off, c0, c1[, chan_scale]): afterwards multiply channels [c0, c1) of the window by the
        LeakyReLU derivative taken from channels [y_off, ...) of ``y`` - and by the Dropout3d keep factors
        ``chan_scale`` [B][c1-c0] when given (the mask then spans the whole window) - folded into the tile
        kernel's epilogue; a separate pass on the generic path."""
        B = g.shape[0]
        cin = s.cin if dx_planar else self.cp(s.cin)
        d = self._desc(s, B, tuple(in_xyz), s.cin if dx_planar else dx.shape[-1], 0 if dx_planar else dx_off,
                       g.shape[-1], g_off, cin=cin, cout=self.cp(s.cout))

        def run():
            m = None if mask is None else (mask[0], mask[1], mask[2], mask[3], self.slope) + tuple(mask[4:5])
            if self.tile_ok(s) and ops.conv_dgrad_tile(d, g, self.filters.get_frag(s.weight, True), dx, alpha=alpha,
                                                       accumulate=accumulate, dx_planar=dx_planar, mask=m,
                                                       acc_src=acc_src, acc_beta=acc_beta, res2=res2, dx2=dx2,
                                                       dx2_c0=dx2_c0):
                return
            if dx2 is not None:
                raise RuntimeError("a two-tensor input gradient needs the tile kernels (set WSR_SPLIT_CAT=0)")
            if acc_src is not None or res2 is not None or acc_beta != 1.0:
                raise RuntimeError("accumulating from another buffer needs the tile kernels (set WSR_WGRAD_STREAM=0)")
            if int(accumulate) > 1:
                raise RuntimeError("partial accumulation needs the tile kernels (set WSR_GD_INPLACE=0)")
            ops.conv_dgrad(d, g, self._wt(s), dx, alpha=alpha, accumulate=accumulate, dx_planar=dx_planar)
            if mask is not None:
                ops.lrelu_bwd_(dx, dx_off + mask[2], mask[0], mask[1], mask[3] - mask[2], self.slope,
                               chan_scale=mask[4] if len(mask) > 4 else None)

        if self.launch_probe is not None:
            self.launch_probe("dgrad:" + s.name, run)
        else:
            run()

    def wgrad(self, s: ConvSite, x: Tensor, x_off: int, g: Tensor, g_off: int, flat: Tensor, space: GradSpace,
              scratch: Tensor, scale: float = 1.0, dst: Optional[Tensor] = None, x2: Optional[Tensor] = None,
              x2_c0: int = 0) -> None:
        """master-layout gradient slot of s.weight (or ``dst``) = scale * wgrad(x[window], g[window]);
        ``x2``: the conv's input channels >= ``x2_c0`` live in this second tensor (see SPLIT_CAT)"""
        B = x.shape[0]
        cin_p = self.cp(s.cin)
        d = self._desc(s, B, tuple(x.shape[1:4]), x.shape[-1], x_off, g.shape[-1], g_off, cin=cin_p)
        out = space.view(flat, s.weight) if dst is None else dst
        if x2 is not None and not DETERMINISTIC:
            raise RuntimeError("a two-tensor filter gradient needs the deterministic form (set WSR_SPLIT_CAT=0)")
        if DETERMINISTIC:
            # (the split count is a property of the un-split conv's geometry: asked with the concatenated width)
            dq = d if x2 is None else self._desc(s, B, tuple(x.shape[1:4]), cin_p, 0, g.shape[-1], g_off, cin=cin_p)
            n = self._wgrad_nparts(("w", s.name, B) + tuple(x.shape[1:4]), dq)
            parts = self._arena_take(n * s.cout * s.taps * cin_p, x.device).view(n, s.cout, s.taps, cin_p)
            run = lambda: ops.conv_wgrad_parts(d, x, g, parts, n, x2=x2, x2_c0=x2_c0)  # noqa: E731
            self._pending_unpack.append((parts[0], out, scale, n, parts[0].numel()))
        else:
            dwp = self._arena_take(s.cout * s.taps * cin_p, x.device)
            run = lambda: ops.conv_wgrad(d, x, g, dwp)  # noqa: E731
            self._pending_unpack.append((dwp.view(s.cout, s.taps, cin_p), out, scale))
        if self.launch_probe is not None:
            self.launch_probe("wgrad:" + s.name, run)
        else:
            run()

    def wgrad_dense(self, convs: Sequence[ConvSite], buf: Tensor, gd: Tensor, flat: Tensor, space: GradSpace,
                    scratch: Tensor) -> None:
        """Filter gradients of all growth convs of one dense block.  bf16: ONE stacked launch
        (``wsr_conv3d_wgrad_tri``) - conv i reads channels [0, nf + i*gc) of ``buf`` and its output
        gradient sits in channels [nf + i*gc, nf + (i+1)*gc) of ``gd``; fp32: one launch per conv."""
        nf, gc = convs[0].cin, convs[0].cout
        stacked = (self.dt == torch.bfloat16 and len(convs) > 1 and all(
            c.kernel == convs[0].kernel and c.stride == (1, 1, 1) and c.pad == convs[0].pad and c.cout == gc
            and c.cin == nf + i * gc and not c.upsample for i, c in enumerate(convs)))
        if not stacked:
            for i, c in enumerate(convs):
                self.wgrad(c, buf, 0, gd, nf + i * gc, flat, space, scratch)
            return
        B = buf.shape[0]
        cin_w, cout = convs[-1].cin, gc * len(convs)
        g = ConvGeom(cin_w, cout, convs[0].kernel, (1, 1, 1), convs[0].pad)
        d = ops.make_desc(g, self.dt, B, tuple(buf.shape[1:4]), buf.shape[-1], 0, gd.shape[-1], nf)
        taps = convs[0].taps
        if DETERMINISTIC:
            n = self._wgrad_nparts(("tri", convs[0].name, B) + tuple(buf.shape[1:4]), d, nf, gc)
            parts = self._arena_take(n * cout * taps * cin_w, buf.device).view(n, cout, taps, cin_w)
            run = lambda: ops.conv_wgrad_parts(d, buf, gd, parts, n, nf, gc)  # noqa: E731
            for i, c in enumerate(convs):
                self._pending_unpack.append((parts[0, i * gc:(i + 1) * gc], space.view(flat, c.weight), 1.0, n,
                                             parts[0].numel()))
        else:
            dw3 = self._arena_take(cout * taps * cin_w, buf.device).view(cout, taps, cin_w)
            run = lambda: ops.conv_wgrad_tri(d, buf, gd, dw3, nf, gc)  # noqa: E731
            for i, c in enumerate(convs):
                self._pending_unpack.append((dw3[i * gc:(i + 1) * gc], space.view(flat, c.weight), 1.0))
        if self.launch_probe is not None:
            self.launch_probe("wgrad_tri:" + convs[0].name, run)
        else:
            run()

    # ---- stacked input gradient of a dense block ---------------------------------------------------
    # Conv i of a block reads channels [0, nf + i*gc) of the dense buffer, so per-conv input gradients
    # read-modify-write nf + i*gc channels each (nf*nc + gc*nc*(nc-1)/2 in total).  Grouped by PRODUCED
    # window instead - window w = the output of conv w-1 (w = 0: the block input) - every window is
    # written once, by one conv whose reduction runs over the output gradients of all convs i >= w,
    # which sit side by side in channels [nf + w*gc, nf + nc*gc) of the gradient buffer: the same
    # arithmetic, a third of the traffic, longer reductions.  Windows go last to first because window
    # w's result (after its LeakyReLU mask) is the output gradient of conv w-1.
    def dense_stackable(self, convs: Sequence[ConvSite]) -> bool:
        if not (self.tile_dt() and len(convs) > 1):
            return False
        nf, gc = convs[0].cin, convs[0].cout
        return (gc % 16 == 0 and nf % 16 == 0 and gc <= 64 and nf <= 256 and convs[0].taps > 1 and all(
            c.kernel == convs[0].kernel and c.stride == (1, 1, 1) and c.pad == convs[0].pad and c.cout == gc
            and c.cin == nf + i * gc and not c.upsample for i, c in enumerate(convs)))

    @staticmethod
    def dense_windows(convs: Sequence[ConvSite]):
        """[(w, c_lo, c_n, red)] - produced channel window and reduction width of each stacked stage"""
        nf, gc, nc = convs[0].cin, convs[0].cout, len(convs)
        return [(w, 0 if w == 0 else nf + (w - 1) * gc, nf if w == 0 else gc, (nc - w) * gc) for w in range(nc)]

    def dense_dgrad_specs(self, convs: Sequence[ConvSite]):
        gc = convs[0].cout
        return [((id(convs[0].weight), w),
                 [(convs[i].weight, True, c_lo, c_n, (i - w) * gc, 0) for i in range(w, len(convs))],
                 c_n, red) for w, c_lo, c_n, red in self.dense_windows(convs)]

    # ---- split forward of a dense block --------------------------------------------------------------
    # Every growth conv reads the block input (channels [0, nf)); that part of all nc convs is ONE conv with
    # nc*gc outputs - a 512-voxel x 128-channel tile runs at ~1.0 PFLOP/s where the 32-output kernels reach
    # ~0.75 - whose raw sums land in the growth windows (bias + LeakyReLU only on conv 0, which has no other
    # input).  Conv i >= 1 then only adds its reduction over the growth channels [nf, nf + i*gc) and applies
    # bias + LeakyReLU after the sum (`act = 2`).  The partial sums pass through bf16 once.
    def dense_fwd_specs(self, convs: Sequence[ConvSite]):
        nf, gc, nc = convs[0].cin, convs[0].cout, len(convs)
        specs = [((id(convs[0].weight), "pre"), [(c.weight, False, 0, nf, 0, i * gc) for i, c in enumerate(convs)],
                  nc * gc, nf)]
        specs += [((id(convs[0].weight), "grow", i), [(convs[i].weight, False, nf, i * gc, 0, 0)], gc, i * gc)
                  for i in range(1, nc)]
        if FWD_REGROUP and self.dt == torch.bfloat16:
            # by source window j (= the output of conv j - 1, channels [nf + (j-1) gc, nf + j gc)): the rows of convs j .. nc-1
            specs += [((id(convs[0].weight), "src", j),
                       [(convs[i].weight, False, nf + (j - 1) * gc, gc, 0, (i - j) * gc) for i in range(j, nc)],
                       (nc - j) * gc, gc) for j in range(1, nc)]
        return specs

    def conv_dense(self, convs: Sequence[ConvSite], buf: Tensor) -> bool:
        """all growth convs of a block on ``buf`` (block input in channels [0, nf)); False = nothing was
        launched (shape outside the tile kernels) and the caller runs the convs one by one"""
        nf, gc, nc = convs[0].cin, convs[0].cout, len(convs)
        B, ctot = buf.shape[0], buf.shape[-1]
        xyz = tuple(buf.shape[1:4])
        k, pad, sl = convs[0].kernel, convs[0].pad, self.slope
        d = ops.make_desc(ConvGeom(nf, nc * gc, k, (1, 1, 1), pad), self.dt, B, xyz, ctot, 0, ctot, nf)
        b0 = convs[0].bias.detach() if convs[0].bias is not None else None
        pre = self.filters.get_stacked((id(convs[0].weight), "pre"))
        done = []

        def run_pre():
            done.append(ops.conv_fwd_tile(d, buf, pre, buf, bias=b0, act=True, slope=sl, act_c1=gc))

        if self.launch_probe is not None:
            self.launch_probe("fwd_dense_pre:" + convs[0].name, run_pre)
        else:
            run_pre()
        if not done[0]:
            return False
        # Grouped by SOURCE window (FWD_REGROUP; volumes of >= 128 tiles of 512 voxels - below that the 128-voxel-tile
        # kernels run and a launch is one workgroup's dependent K-loop either way): stage j adds the contribution of window
        # j to the windows of convs j .. nc-1 and completes conv j's (bias + LeakyReLU on its first gc channels only).  Each
        # later window's partial sums pass through bf16 once per stage (conv 3: three times instead of once).
        if (FWD_REGROUP and self.dt == torch.bfloat16 and nc > 2 and gc * (nc - 1) <= 96
                and B * xyz[0] * xyz[1] * xyz[2] >= 128 * 512):
            for j in range(1, nc):
                n_out = (nc - j) * gc
                dj = ops.make_desc(ConvGeom(gc, n_out, k, (1, 1, 1), pad), self.dt, B, xyz, ctot, nf + (j - 1) * gc, ctot,
                                   nf + j * gc)
                bj = convs[j].bias.detach() if convs[j].bias is not None else None
                fr = self.filters.get_stacked((id(convs[0].weight), "src", j))
                ok = []

                def run_src(dj=dj, bj=bj, fr=fr, off=nf + j * gc, c1=gc if j < nc - 1 else 0):
                    ok.append(ops.conv_fwd_tile(dj, buf, fr, buf, bias=bj, res=buf, res_off=off, beta=1.0, act=2, slope=sl,
                                                act_c1=c1))

                if self.launch_probe is not None:
                    self.launch_probe(f"fwd_dense_src{j}:" + convs[0].name, run_src)
                else:
                    run_src()
                if not ok[0]:
                    if j == 1:
                        break  # nothing written yet: the per-conv stages below take over
                    raise RuntimeError("source-grouped dense-block stage outside the tile kernels (set WSR_FWD_REGROUP=0)")
            else:
                return True
        for i in range(1, nc):
            di = ops.make_desc(ConvGeom(i * gc, gc, k, (1, 1, 1), pad), self.dt, B, xyz, ctot, nf, ctot, nf + i * gc)
            bi = convs[i].bias.detach() if convs[i].bias is not None else None
            fr = self.filters.get_stacked((id(convs[0].weight), "grow", i))

            def run(di=di, bi=bi, fr=fr, off=nf + i * gc):
                if not ops.conv_fwd_tile(di, buf, fr, buf, bias=bi, res=buf, res_off=off, beta=1.0, act=2, slope=sl):
                    raise RuntimeError("split dense-block conv: second stage outside the tile kernels "
                                       "(set WSR_STACK_FWD=0)")

            if self.launch_probe is not None:
                self.launch_probe(f"fwd_dense_grow{i}:" + convs[0].name, run)
            else:
                run()
        return True

    def dgrad_dense(self, convs: Sequence[ConvSite], gd: Tensor, buf: Tensor, in_xyz) -> None:
        """gd[..., :nf + (nc-1)*gc] += input gradients of all growth convs (gd[..., nf + i*gc:][:gc] = output
        gradient of conv i, complete for i = nc-1 on entry); LeakyReLU derivative of ``buf`` applied to the
        growth windows as they become final."""
        nf, gc, nc = convs[0].cin, convs[0].cout, len(convs)
        B, ctot = gd.shape[0], gd.shape[-1]
        for w, c_lo, c_n, red in reversed(self.dense_windows(convs)):
            geom = ConvGeom(c_n, red, convs[0].kernel, (1, 1, 1), convs[0].pad)
            d = ops.make_desc(geom, self.dt, B, tuple(in_xyz), ctot, c_lo, ctot, nf + w * gc)
            m = None if w == 0 else (buf, c_lo, 0, c_n, self.slope)
            frag = self.filters.get_stacked((id(convs[0].weight), w))

            def run():
                if ops.conv_dgrad_tile(d, gd, frag, gd, alpha=1.0, accumulate=True, mask=m):
                    return
                for i in range(w, nc):  # generic kernels, one source conv at a time
                    c = convs[i]
                    di = self._desc(c, B, tuple(in_xyz), ctot, c_lo, ctot, nf + i * gc, cin=c_n, cout=self.cp(c.cout))
                    ops.conv_dgrad(di, gd, self._wt(c)[c_lo:c_lo + c_n], gd, accumulate=True)
                if m is not None:
                    ops.lrelu_bwd_(gd, c_lo, buf, c_lo, c_n, self.slope)

            if self.launch_probe is not None:
                self.launch_probe(f"dgrad_dense{w}:" + convs[0].name, run)
            else:
                run()

    # ---- packed filter-gradient arena -------------------------------------------------------------
    # The wgrad kernels accumulate (float atomics) into packed [Cout][taps][Cin_p] buffers.  One arena per
    # backward pass is zeroed with a single fill, every conv takes a slice, and the slices are moved to
    # the master layout in batches (one launch per `flush_unpack`) instead of a fill + an unpack per conv.
    def begin_backward(self, dev) -> None:
        self._pending_unpack = []
        self._arena_off = 0
        self._arena_need = 0
        self._arena_dev = dev
        self._events_used = 0
        if DETERMINISTIC:
            # split copies are written with plain stores: nothing to zero.  The arena is PERSISTENT (sized by the
            # first backward pass of a shape): stable addresses keep the cached unpack job tables valid
            if self._arena is not None and self._arena.device != torch.device(dev):
                self._arena = None
            return
        n = int(self._scratch_elems_total * 1.3) + 4096
        self._arena = torch.zeros(n, dtype=torch.float32, device=dev)

    def _wgrad_nparts(self, key, desc, tri_base: int = 0, tri_step: int = 0) -> int:
        if self._env_gen != ops.ENV_GEN[0]:  # the tuning switches were re-read: plans and job tables start over
            self._nparts.clear()
            self._unpack_tables.clear()
            self._env_gen = ops.ENV_GEN[0]
        n = self._nparts.get(key)
        if n is None:
            n = self._nparts[key] = ops.conv_wgrad_nparts(desc, tri_base, tri_step)
        return n

    def _arena_reserve(self, n_total: int, dev) -> None:
        """make sure the next slices of ``n_total`` floats in all fit WITHOUT a recycling flush in between (call sites
        that take several slices first and launch their kernels afterwards: a flush in the middle would reduce copies
        that have not been written yet)"""
        if not DETERMINISTIC:
            return
        need = n_total + 64 * 64
        bound = max(ARENA_MB << 18, need)
        if self._arena is not None and self._arena.device != torch.device(dev):
            self.flush_unpack()
            self._arena = None
        have = 0 if self._arena is None else self._arena.numel()
        if self._arena_off + need > have and have < bound:
            want = min(bound, max(2 * have, self._arena_off + need, 4 << 18))
            self.flush_unpack()
            self._arena = None
            self._arena = torch.empty(want, dtype=torch.float32, device=dev)
            self._arena_off = 0
            self._unpack_tables.clear()
        elif self._arena_off + need > self._arena.numel():
            self.flush_unpack()
            self._arena_off = 0

    def _arena_take(self, n: int, dev) -> Tensor:
        if DETERMINISTIC:
            # bounded, persistent arena: when the next slice does not fit, everything taken so far is reduced into the
            # master gradients and the arena is recycled (stream order keeps the reduce ahead of the next writer).  The
            # sequence of slices is the same every step, so the cached device job tables stay valid.
            # ... It starts at what the first slices need and doubles up to the bound (ARENA_MB) as a pass asks for more:
            # a small-patch model (a few MB of copies per pass) no longer pins the full bound from its first backward.
            bound = max(ARENA_MB << 18, (n + 63) // 64 * 64)  # (floats)
            if self._arena is not None and self._arena.device != torch.device(dev):
                self.flush_unpack()
                self._arena = None
            have = 0 if self._arena is None else self._arena.numel()
            if self._arena_off + n > have and have < bound:
                want = min(bound, max(2 * have, self._arena_off + (n + 63) // 64 * 64, 4 << 18))
                self.flush_unpack()
                self._arena = None  # (release before growing)
                self._arena = torch.empty(want, dtype=torch.float32, device=dev)
                self._arena_off = 0
                self._unpack_tables.clear()  # (cached job tables hold the old addresses)
            if self._arena_off + n > self._arena.numel():
                self.flush_unpack()
                self._arena_off = 0
            off = self._arena_off
            self._arena_off = off + (n + 63) // 64 * 64
            return self._arena[off:off + n]
        off = self._arena_off
        self._arena_need += (n + 63) // 64 * 64
        if self._arena is None or off + n > self._arena.numel():
            return torch.zeros(n, dtype=torch.float32, device=dev)
        self._arena_off = off + (n + 63) // 64 * 64
        return self._arena[off:off + n]

    def flush_unpack(self) -> None:
        if not self._pending_unpack:
            return
        jobs = self._pending_unpack
        self._pending_unpack = []
        # the arena and the flat gradient buffer usually come back at the same addresses every step
        # (caching allocator), so the device job table is re-used when all pointers match
        key = tuple(j[0].data_ptr() for j in jobs) + tuple(j[1].data_ptr() for j in jobs) + tuple(j[2:] for j in jobs)
        table = self._unpack_tables.get(key)
        if table is None:
            if len(self._unpack_tables) > 512:
                self._unpack_tables.clear()
            table = ops.unpack_job_table(jobs)
            self._unpack_tables[key] = table
        if len(jobs[0]) > 3:
            ops.unpack_wgrad_reduce_multi(table)  # ordered sum over the split copies + move to the master layout
        else:
            ops.unpack_wgrad_multi(table)

    def end_backward(self) -> None:
        self.flush_unpack()
        if not DETERMINISTIC:
            self._arena = None

    def release_buffers(self) -> None:
        """give the filter-gradient arena back (a model that leaves training for good; it is re-made on demand)"""
        self.flush_unpack()
        self._arena = None
        self._unpack_tables.clear()

    @staticmethod
    def wgrad_scratch_elems(sites: Sequence[ConvSite], e: int) -> int:
        return max(s.cout * s.taps * ((s.cin + e - 1) // e * e) for s in sites)

    def _empty(self, shape, like: Tensor, zero: bool = False) -> Tensor:
        if zero:
            return torch.zeros(shape, dtype=self.dt, device=like.device)
        if POISON_BUFFERS:  # tests: every element must be written by a kernel before it is read
            return torch.full(shape, float("nan"), dtype=self.dt, device=like.device)
        return torch.empty(shape, dtype=self.dt, device=like.device)


# =============================================================================
# Generator
# =============================================================================
class GeneratorProgram(ProgramBase):
    """Forward / backward of ``Generator_3D`` (reference Generator_3D_Resnet_ESRGAN.py:225-229)."""

    def __init__(self, G: nn.Module, dt: torch.dtype):
        super().__init__(dt)
        m = G.model
        self.slope = G.slope
        self.dropout_p = G.hr_convs[1].p
        self.feature = site_from_conv("model.0.0", m[0][0])
        trunk = m[1].module
        self.rrdbs: List[List[Tuple[List[ConvSite], ConvSite, float]]] = []
        self.rrdb_scales: List[float] = []
        n_rrdb = len(trunk) - 1
        for r in range(n_rrdb):
            rr = trunk[r]
            rdbs = []
            for d, rdb in enumerate(rr.RDBs):
                convs = [site_from_conv(f"model.1.module.{r}.RDBs.{d}.conv{i}.conv.0", getattr(rdb, f"conv{i}").conv[0])
                         for i in range(rdb.number_of_convs)]
                lff = site_from_conv(f"model.1.module.{r}.RDBs.{d}.LFF", rdb.LFF)
                rdbs.append((convs, lff, float(rdb.residual_scaling)))
            self.rrdbs.append(rdbs)
            self.rrdb_scales.append(float(rr.RRDB_residual_scaling))
        self.lr_conv = site_from_conv(f"model.1.module.{n_rrdb}.0", trunk[n_rrdb][0])
        self.ups = [site_from_conv(f"model.{2 + u}.1.0", m[2 + u][1][0], upsample=True) for u in range(len(m) - 2)]
        self.terrain0 = site_from_conv("terrain_convs.0.0", G.terrain_convs[0][0])
        self.terrain1 = site_from_conv("terrain_convs.1.0", G.terrain_convs[1][0])
        self.hr0 = site_from_conv("hr_convs.0.0", G.hr_convs[0][0])
        self.hr1 = site_from_conv("hr_convs.2", G.hr_convs[2])
        # z-folded twin of hr1 (see wsr_zfold): a (KX,KY,1) conv with cout*KZ outputs whose filter is a
        # permuted copy of hr1's, refreshed when that changes
        self.hr1z: Optional[ConvSite] = None
        self._hr1z_stamp = None
        self._hr1z_grad: Optional[Tensor] = None
        k = self.hr1.kernel
        if k[2] > 1 and self.hr1.cout * k[2] <= 16 and self.hr1.stride == (1, 1, 1) and not self.hr1.upsample:
            w = self.hr1.weight
            wz = torch.empty((self.hr1.cout * k[2], self.hr1.cin, k[0], k[1], 1), dtype=torch.float32, device=w.device)
            self.hr1z = ConvSite("hr_convs.2.zfold", wz, None, (k[0], k[1], 1), (1, 1, 1),
                                 (self.hr1.pad[0], self.hr1.pad[1], 0))
        # ... and its filter gradient with the roles of the two operands exchanged:
        #   dW[n, t, c] = sum_v dy[v, n] x[v + t - p, c] = sum_u x[u, c] dy[u + (K-1-t) - p, n]     (p = (K-1)/2)
        # is the filter gradient of a conv 16 -> 144 whose "input" is dy and whose "output gradient" is x, taps flipped.
        # As stated the 144-channel activation is the halo image, fetched (K+7)^2/64 = 2.25 times per chunk of 32 of its
        # channels, against ONE 16-wide n-tile: two transposing reads per MFMA.  Exchanged, the activation is read once, the
        # 16-channel image carries the halo and every fragment meets three or four others.
        self._hr1z_t: Optional[ConvSite] = None
        self._hr1z_grad_t: Optional[Tensor] = None
        if self.hr1z is not None and SWAP_THIN_WGRAD and k[0] % 2 == 1 and k[1] % 2 == 1 and \
                self.hr1.pad[:2] == (k[0] // 2, k[1] // 2):
            wt = torch.empty((self.hr1.cin, (self.hr1.cout * k[2] + 7) // 8 * 8, k[0], k[1], 1), device="meta")
            self._hr1z_t = ConvSite("hr_convs.2.zfold.T", wt, None, (k[0], k[1], 1), (1, 1, 1),
                                    (self.hr1.pad[0], self.hr1.pad[1], 0))
        # Sub-pixel form of the up-sampling convs (reference torch_blocks.py:345-347: nn.Upsample(scale_factor=(2,2,1),
        # mode="nearest") in front of a 3x3x3 conv): output parity (a, b) only ever sees 2x2 distinct un-sampled
        # voxels per z level, so the conv is four 2x2x3 convs on the un-sampled input whose filters are sums of the
        # master taps (wsr_subpixel_fold) - 12 instead of 27 taps per output voxel.  Parity sites are twins of the
        # master filter like hr1z; they serve the tile kernels (bf16; fp32: forward and input gradient).
        self.up_parity: List[Optional[List[ConvSite]]] = []
        self._up_wp: List[Optional[Tensor]] = []
        self._up_dwp: List[Optional[Tensor]] = []  # parity filter gradients (folded back by wsr_subpixel_unfold)
        self._up_stamp: List[object] = []
        for u in self.ups:
            ok = (u.kernel[0], u.kernel[1]) == (3, 3) and (u.pad[0], u.pad[1]) == (1, 1) and u.stride == (1, 1, 1)
            if ok:
                kz = u.kernel[2]
                wp = torch.empty((4, u.cout, u.cin, 2, 2, kz), dtype=torch.float32, device=u.weight.device)
                self.up_parity.append([ConvSite(f"{u.name}.parity{ph}", wp[ph], None, (2, 2, kz), (1, 1, 1),
                                                (1 - (ph >> 1), 1 - (ph & 1), u.pad[2])) for ph in range(4)])
            else:
                wp = None
                self.up_parity.append(None)
            self._up_wp.append(wp)
            self._up_dwp.append(None)
            self._up_stamp.append(None)
        self.nf = self.feature.cout
        self.gc = self.rrdbs[0][0][0][0].cout if self.rrdbs and self.rrdbs[0][0][0] else 0
        self.tf = self.terrain1.cout
        if self.nf % self.e or (self.gc % self.e):
            raise ValueError(f"num_features ({self.nf}) and RDB_growth_chan ({self.gc}) must be multiples of "
                             f"{self.e} for compute dtype {dt}")
        # backward production order (first produced first) for the flat gradient space
        order: List[nn.Parameter] = [self.hr1.weight, self.hr1.bias, self.hr0.weight, self.terrain1.weight,
                                     self.terrain0.weight]
        order += [u.weight for u in reversed(self.ups)]
        order.append(self.lr_conv.weight)
        for rdbs in reversed(self.rrdbs):
            for convs, lff, _ in reversed(rdbs):
                order += [lff.weight, lff.bias] + [c.weight for c in reversed(convs)]
        order.append(self.feature.weight)
        self.space = GradSpace(order)
        self.param_list = order
        self.all_sites = ([self.feature, self.lr_conv, self.terrain0, self.terrain1, self.hr0, self.hr1] + self.ups
                          + [c for rdbs in self.rrdbs for convs, lff, _ in rdbs for c in convs + [lff]])
        if self.hr1z is not None:
            self.all_sites.append(self.hr1z)
        self._scratch_elems = self.wgrad_scratch_elems(self.all_sites, self.e)
        if self.rrdbs and self.rrdbs[0][0][0]:
            c = self.rrdbs[0][0][0]
            self._scratch_elems = max(self._scratch_elems, len(c) * c[0].cout * c[0].taps * self.cp(c[-1].cin))
        self._scratch_elems_total = sum(s.cout * s.taps * self.cp(s.cin) for s in self.all_sites)

    def zfold_active(self) -> bool:
        return ZFOLD and self.hr1z is not None and self.tile_ok(self.hr1z)

    def subpixel_active(self, u: int) -> bool:
        return (SUBPIXEL and self.up_parity[u] is not None and self.tile_dt()
                and self.cp(self.ups[u].cin) == self.ups[u].cin)

    def conv_sites(self) -> Sequence[ConvSite]:
        zf = self.zfold_active()
        sites = [s for s in self.all_sites if s is not (self.hr1 if zf else self.hr1z)]
        for u, par in enumerate(self.up_parity):
            if self.subpixel_active(u):  # the parity twins are what the tile kernels read; the master is not packed
                sites = [s for s in sites if s is not self.ups[u]] + par
        return sites

    def _refresh_parity(self, u: int) -> None:
        """parity filters of up-conv u from its master filter, when that changed"""
        site, par = self.ups[u], self.up_parity[u]
        w = site.weight
        wp = self._up_wp[u]
        if wp.device != w.device:
            wp = self._up_wp[u] = torch.empty_like(wp, device=w.device)
            for ph, s in enumerate(par):
                s.weight = wp[ph]
            self._up_stamp[u] = None
        # the four forward fragment filters share one buffer, parity-major: one launch runs all parities
        if self._seed_parity_frags(par, w.device):
            self._up_stamp[u] = None
        stamp = (w._version, w.data_ptr(), self.filters._gen)
        if stamp != self._up_stamp[u]:
            ops.subpixel_fold(w.detach().contiguous(), wp)
            self.filters.touch()
            self._up_stamp[u] = stamp

    def refresh_filters(self, backward: bool) -> None:
        if self.zfold_active():  # hr1's filter in the folded arrangement [c*KZ + kz][ci][kx][ky][0]
            w = self.hr1.weight
            wz = self.hr1z.weight
            if wz.device != w.device:
                wz = self.hr1z.weight = torch.empty_like(wz, device=w.device)
                self._hr1z_stamp = None
            stamp = (w._version, w.data_ptr(), self.filters._gen)
            if stamp != self._hr1z_stamp:
                kx, ky, kz = self.hr1.kernel
                wz.view(self.hr1.cout, kz, self.hr1.cin, kx, ky).copy_(w.detach().permute(0, 4, 1, 2, 3))
                self._hr1z_stamp = stamp
        for u in range(len(self.ups)):
            if self.subpixel_active(u):
                self._refresh_parity(u)
        super().refresh_filters(backward)

    def up_conv(self, u: int, cur: Tensor, out: Tensor) -> None:
        """up-conv u: nearest x(2,2,1) + conv + LeakyReLU, ``cur`` (B, X, Y, Z, nf) -> window [0, cout) of ``out``
        (B, 2X, 2Y, Z, .)"""
        site, sl = self.ups[u], self.slope
        if not self.subpixel_active(u):
            self.conv(site, cur, 0, out, 0, act=True, slope=sl)
            return
        par = self.up_parity[u]
        B, xyz = cur.shape[0], tuple(cur.shape[1:4])
        bias = site.bias.detach() if site.bias is not None else None
        frs, batched = self._parity_frags(par)

        def run():
            if batched:  # one launch, parity = two bits of the workgroup index
                d = ops.make_desc(ConvGeom(site.cin, site.cout, par[0].kernel, (1, 1, 1), par[0].pad), self.dt, B, xyz,
                                  cur.shape[-1], 0, out.shape[-1], 0, lat=(0, 0, 4))
                if ops.conv_fwd_tile(d, cur, frs[0], out, bias=bias, act=True, slope=sl):
                    return
            for ph, s in enumerate(par):
                d = ops.make_desc(ConvGeom(site.cin, site.cout, s.kernel, (1, 1, 1), s.pad), self.dt, B, xyz,
                                  cur.shape[-1], 0, out.shape[-1], 0, lat=(ph >> 1, ph & 1, 0))
                if not ops.conv_fwd_tile(d, cur, frs[ph], out, bias=bias, act=True, slope=sl):
                    raise RuntimeError("sub-pixel up-conv outside the tile kernels (set WSR_SUBPIXEL=0)")

        if self.launch_probe is not None:
            self.launch_probe("fwd:" + site.name, run)
        else:
            run()

    def up_wgrad(self, u: int, inp: Tensor, g: Tensor, flat: Tensor, space: "GradSpace") -> None:
        """filter gradient of up-conv u in its sub-pixel form: per parity the 2x2x3 gradient over the un-sampled
        input and that parity's lattice of the output gradient (12 instead of 27 taps per output voxel), folded
        back onto the 3x3x3 master filter by the adjoint of the tap sums (``wsr_subpixel_unfold``)."""
        site, par = self.ups[u], self.up_parity[u]
        B, xyz = inp.shape[0], tuple(inp.shape[1:4])
        dwp = self._up_dwp[u]
        if dwp is None or dwp.device != inp.device:
            dwp = self._up_dwp[u] = torch.empty_like(self._up_wp[u], device=inp.device)
        cin_p = self.cp(site.cin)
        runs = []
        descs = [ops.make_desc(ConvGeom(site.cin, site.cout, s.kernel, (1, 1, 1), s.pad), self.dt, B, xyz, inp.shape[-1],
                               0, g.shape[-1], 0, cin=cin_p, lat=(ph >> 1, ph & 1, 0)) for ph, s in enumerate(par)]
        if DETERMINISTIC:  # (the launches below are deferred: no recycling flush between the slices)
            self._arena_reserve(sum((self._wgrad_nparts(("wpar", site.name, ph, B) + xyz, d) * site.cout * s.taps * cin_p
                                     + 63) // 64 * 64 for ph, (s, d) in enumerate(zip(par, descs))), inp.device)
        for ph, s in enumerate(par):
            d = descs[ph]
            if DETERMINISTIC:
                n = self._wgrad_nparts(("wpar", site.name, ph, B) + xyz, d)
                parts = self._arena_take(n * site.cout * s.taps * cin_p, inp.device).view(n, site.cout, s.taps, cin_p)
                runs.append(lambda d=d, parts=parts, n=n: ops.conv_wgrad_parts(d, inp, g, parts, n))
                self._pending_unpack.append((parts[0], dwp[ph], 1.0, n, parts[0].numel()))
            else:
                dw = self._arena_take(site.cout * s.taps * cin_p, inp.device)
                runs.append(lambda d=d, dw=dw: ops.conv_wgrad(d, inp, g, dw))
                self._pending_unpack.append((dw.view(site.cout, s.taps, cin_p), dwp[ph], 1.0))

        def run():
            for r in runs:
                r()

        if self.launch_probe is not None:
            self.launch_probe("wgrad:" + site.name, run)
        else:
            run()
        self.flush_unpack()
        ops.subpixel_unfold(dwp, space.view(flat, site.weight))

    def up_dgrad(self, u: int, g: Tensor, gin: Tensor, mask=None) -> None:
        """input gradient of up-conv u in its sub-pixel form: ``gin`` (B, X, Y, Z, nf) = sum over the four parities of
        the 2x2x3 input gradient of the output gradient on that parity's lattice of ``g`` (B, 2X, 2Y, Z, .)"""
        site, par = self.ups[u], self.up_parity[u]
        B, xyz = g.shape[0], tuple(gin.shape[1:4])

        def run():
            for ph, s in enumerate(par):
                d = ops.make_desc(ConvGeom(site.cin, site.cout, s.kernel, (1, 1, 1), s.pad), self.dt, B, xyz,
                                  gin.shape[-1], 0, g.shape[-1], 0, cin=self.cp(site.cin), cout=self.cp(site.cout),
                                  lat=(ph >> 1, ph & 1, 0))
                m = None if (mask is None or ph < 3) else (mask[0], mask[1], mask[2], mask[3], self.slope)
                if not ops.conv_dgrad_tile(d, g, self.filters.get_frag(s.weight, True), gin, accumulate=ph > 0, mask=m):
                    raise RuntimeError("sub-pixel up-conv gradient outside the tile kernels (set WSR_SUBPIXEL=0)")

        if self.launch_probe is not None:
            self.launch_probe("dgrad:" + site.name, run)
        else:
            run()

    def _trunk(self, first: Tensor, x: Tensor, bufs: List[Tensor]) -> Tensor:
        """the RRDB stack on ``first`` (block input in channels [0, nf) of a dense buffer); appends every block's dense
        buffer to ``bufs`` (the backward pass reads them) and returns the stack's output (nf channels)"""
        nf, gc, sl = self.nf, self.gc, self.slope
        B, X, Y, nz = first.shape[:4]
        nconv = len(self.rrdbs[0][0][0]) if self.rrdbs else 0
        dense = nf + nconv * gc
        total_rdbs = sum(len(r) for r in self.rrdbs)
        buf = first
        seen = 0
        for rdbs, rr_scale in zip(self.rrdbs, self.rrdb_scales):
            rr_in = buf
            for convs, lff, rdb_scale in rdbs:
                if not (STACK_FWD and self.dense_stackable(convs) and self.conv_dense(convs, buf)):
                    for i, c in enumerate(convs):
                        self.conv(c, buf, 0, buf, nf + i * gc, act=True, slope=sl)
                seen += 1
                nb = self._empty((B, X, Y, nz, nf if seen == total_rdbs else dense), x)
                # x + rdb_scale * (LFF(dense) + b); the last block of an RRDB also takes the RRDB shortcut:
                # rr_scale * (x + rdb_scale * (LFF + b)) + x_rr in the same launch (two residuals)
                fold = (convs is rdbs[-1][0] and self.dt == torch.bfloat16 and self.use_tile
                        and ops.conv1x1_covers(buf.shape[-1], nf, False) and lff.taps == 1)
                if fold:
                    self.conv(lff, buf, 0, nb, 0, alpha=rr_scale * rdb_scale, res=buf, res_off=0, beta=rr_scale,
                              res2=rr_in, res2_off=0, beta2=1.0)
                else:
                    self.conv(lff, buf, 0, nb, 0, alpha=rdb_scale, res=buf, res_off=0, beta=1.0)
                bufs.append(buf)
                buf = nb
            # RRDB residual: out = rr_scale * chain + x_rr
            if not (rdbs and fold):
                ops.chan_axpby(buf, 0, rr_in, 0, nf, alpha=1.0, beta=rr_scale)
        return buf

    # ---- single stages (inference only) ---------------------------------------------
    def run_stage(self, stack: str, idx: int, x: Tensor) -> Tensor:
        """One element of ``Generator_3D.model`` / ``.terrain_convs`` / ``.hr_convs`` on a planar fp32
        (B, C, X, Y, Z) tensor, as the reference's ``nn.Sequential`` children compute it - what
        ``G.model[:2](LR)``, ``G.terrain_convs(Z)``, ``G.hr_convs[:-2](t)`` are made of (reference
        plot_data.py:770-793).  No gradients, no saved state; the full forward pass stays ``GeneratorProgram.forward``."""
        ops._need_cuda(x, self.feature.weight)
        self.refresh_filters(backward=False)
        x = x.detach().contiguous().float()
        B, C_, X, Y, nz = x.shape
        nf, sl = self.nf, self.slope

        def to_nd(c_fill, ctot=None):
            buf = self._empty((B, X, Y, nz, ctot or c_fill), x)
            ops.planar_to_ndhwc(x, buf, 0, c_fill)
            return buf

        def need(c):
            if C_ != c:
                raise ValueError(f"{stack}[{idx}] takes {c} channels, got {C_}")

        nconv = len(self.rrdbs[0][0][0]) if self.rrdbs else 0
        dense = nf + nconv * self.gc if self.rrdbs else nf
        if stack == "model" and idx == 0:      # feature conv
            need(self.feature.cin)
            y = self._empty((B, X, Y, nz, nf), x)
            self.conv(self.feature, to_nd(self.cp(C_)), 0, y, 0)
            return ops.ndhwc_to_planar(y, nf)
        if stack == "model" and idx == 1:      # x + lr_conv(RRDB stack(x))
            need(nf)
            first = to_nd(nf, dense)
            t_last = self._trunk(first, x, [])
            y = self._empty((B, X, Y, nz, nf), x)
            self.conv(self.lr_conv, t_last, 0, y, 0, res=first, res_off=0, beta=1.0)
            return ops.ndhwc_to_planar(y, nf)
        if stack == "model" and 2 <= idx < 2 + len(self.ups):   # nearest x(2,2,1) + conv + LeakyReLU
            need(nf)
            y = self._empty((B, 2 * X, 2 * Y, nz, nf), x)
            self.up_conv(idx - 2, to_nd(nf), y)
            return ops.ndhwc_to_planar(y, nf)
        if stack == "terrain_convs" and idx in (0, 1):
            site = (self.terrain0, self.terrain1)[idx]
            need(site.cin)
            y = self._empty((B, X, Y, nz, self.cp(site.cout)), x, zero=self.cp(site.cout) != site.cout)
            if idx == 0:
                self.conv(site, to_nd(self.cp(C_)), 0, y, 0, act=True, slope=sl)
            else:
                self.conv(site, to_nd(self.cp(C_)), 0, y, 0)
            return ops.ndhwc_to_planar(y, site.cout)
        if stack == "hr_convs" and idx == 0:   # conv + LeakyReLU (Dropout3d is hr_convs[1], a torch module)
            need(self.hr0.cin)
            y = self._empty((B, X, Y, nz, self.cp(self.hr0.cout)), x, zero=self.cp(self.hr0.cout) != self.hr0.cout)
            self.conv(self.hr0, to_nd(self.cp(C_)), 0, y, 0, act=True, slope=sl)
            return ops.ndhwc_to_planar(y, self.hr0.cout)
        if stack == "hr_convs" and idx == 2:   # last conv (+ bias), planar out
            need(self.hr1.cin)
            h = to_nd(self.cp(C_))
            out = torch.empty((B, self.hr1.cout, X, Y, nz), dtype=torch.float32, device=x.device)
            if self.zfold_active():
                kz = self.hr1.kernel[2]
                parts = torch.empty((B, self.hr1.cout * kz, X, Y, nz), dtype=torch.float32, device=x.device)
                self.conv(self.hr1z, h, 0, parts, 0, out_planar=True)
                ops.zfold(parts, out, self.hr1.bias.detach() if self.hr1.bias is not None else None, kz, self.hr1.pad[2])
            else:
                self.conv(self.hr1, h, 0, out, 0, out_planar=True)
            return out
        raise IndexError(f"Generator_3D.{stack} has no element {idx} that runs on the HIP program")

    # ---- forward -------------------------------------------------------------------
    def forward(self, x: Tensor, Z: Tensor, training: bool, save: bool, drop_scale: Optional[Tensor]):
        """x (B, Cin, X, Y, nz), Z (B, 1, sX, sY, nz) planar fp32 -> (B, 3, sX, sY, nz) fp32 (+ saved state)"""
        B, _, X, Y, nz = x.shape
        nf, gc, tf, sl = self.nf, self.gc, self.tf, self.slope
        ops._need_cuda(x, Z, self.feature.weight)  # inputs and parameters on the current device
        self.refresh_filters(backward=save)
        x = x.contiguous().float()
        Z = Z.contiguous().float()
        cin_p = self.cp(self.feature.cin)
        x_nd = self._empty((B, X, Y, nz, cin_p), x)
        ops.planar_to_ndhwc(x, x_nd, 0, cin_p)
        nconv = len(self.rrdbs[0][0][0]) if self.rrdbs else 0
        dense = nf + nconv * gc
        total_rdbs = sum(len(r) for r in self.rrdbs)
        first = self._empty((B, X, Y, nz, dense if total_rdbs else nf), x)
        self.conv(self.feature, x_nd, 0, first, 0)
        bufs: List[Tensor] = []
        buf = self._trunk(first, x, bufs)
        t_last = buf
        s = self._empty((B, X, Y, nz, nf), x)
        self.conv(self.lr_conv, t_last, 0, s, 0, res=first, res_off=0, beta=1.0)
        sX, sY = X * (2 ** len(self.ups)), Y * (2 ** len(self.ups))
        cat_c = self.cp(nf + tf)
        tf_p = self.cp(tf)
        # the concat as two dense tensors where the 5x5x5 conv's kernels read them side by side (SPLIT_CAT)
        split = bool(SPLIT_CAT and self.ups and DETERMINISTIC and self.tile_ok(self.hr0) and nf + tf == cat_c and tf_p == tf
                     and ops.conv_split_ok(self._desc(self.hr0, B, (sX, sY, nz), nf, 0, cat_c, 0, cin=cat_c), nf))
        hcat = self._empty((B, sX, sY, nz, nf if split else cat_c), x, zero=cat_c != nf + tf)
        tfeat = self._empty((B, sX, sY, nz, tf_p), x) if split else None
        cur = s
        up_io: List[Tuple[Tensor, Tensor]] = []
        for u, site in enumerate(self.ups):
            last = u == len(self.ups) - 1
            out = hcat if last else self._empty((B, cur.shape[1] * 2, cur.shape[2] * 2, nz, nf), x)
            self.up_conv(u, cur, out)
            up_io.append((cur, out))
            cur = out
        if not self.ups:
            ops.chan_axpby(hcat, 0, s, 0, nf)
        z_p = self.cp(1)
        z_nd = self._empty((B, sX, sY, nz, z_p), x)
        ops.planar_to_ndhwc(Z, z_nd, 0, z_p)
        t0 = self._empty((B, sX, sY, nz, tf_p), x, zero=tf_p != tf)
        self.conv(self.terrain0, z_nd, 0, t0, 0, act=True, slope=sl)
        h = self._empty((B, sX, sY, nz, cat_c), x, zero=cat_c != nf + tf)
        if split:
            self.conv(self.terrain1, t0, 0, tfeat, 0)
            self.conv(self.hr0, hcat, 0, h, 0, act=True, slope=sl, chan_scale=drop_scale, in2=tfeat, in2_c0=nf)
        else:
            self.conv(self.terrain1, t0, 0, hcat, nf)
            self.conv(self.hr0, hcat, 0, h, 0, act=True, slope=sl, chan_scale=drop_scale)
        out = torch.empty((B, self.hr1.cout, sX, sY, nz), dtype=torch.float32, device=x.device)
        if self.zfold_active():
            kz = self.hr1.kernel[2]
            parts = torch.empty((B, self.hr1.cout * kz, sX, sY, nz), dtype=torch.float32, device=x.device)
            self.conv(self.hr1z, h, 0, parts, 0, out_planar=True)
            ops.zfold(parts, out, self.hr1.bias.detach() if self.hr1.bias is not None else None, kz, self.hr1.pad[2])
            del parts
        else:
            self.conv(self.hr1, h, 0, out, 0, out_planar=True)
        saved = None
        if save:
            saved = dict(x_nd=x_nd, first=first, bufs=bufs, t_last=t_last, s=s, up_io=up_io, hcat=hcat, tfeat=tfeat, z_nd=z_nd,
                         t0=t0, h=h, drop=drop_scale, lr_xyz=(X, Y, nz), hr_xyz=(sX, sY, nz))
        return out, saved

    # ---- backward ------------------------------------------------------------------
    def backward(self, saved: dict, g_out: Tensor) -> Tensor:
        """g_out (B, 3, sX, sY, nz) fp32 -> flat fp32 gradient buffer (see ``self.space``)."""
        nf, gc, tf, sl = self.nf, self.gc, self.tf, self.slope
        dev = g_out.device
        B = g_out.shape[0]
        X, Y, nz = saved["lr_xyz"]
        sX, sY, _ = saved["hr_xyz"]
        flat = self.space.new(dev)
        scratch = None  # (packed gradients live in the per-backward arena)
        self.begin_backward(dev)
        sp = self.space
        done = 0

        def ready(*params):
            nonlocal done
            if self.grad_ready_hook is None:
                return
            # (the gradients of a bucket must be in the flat buffer before it is reduced: the hook calls flush_unpack
            # right before it launches a collective - flushing at every call cost 50 extra unpack launches per step)
            hi = max(sp.offsets[id(p)][0] + (sp.offsets[id(p)][1] + 63) // 64 * 64 for p in params)
            if hi > done:
                self.grad_ready_hook(flat, done, hi, self.flush_unpack)
                done = hi

        def tr(tag, idx, t):  # (test aid, see ProgramBase.trace)
            if self.trace is not None:
                self.trace.append((tag, idx, t.clone()))

        g_out = g_out.contiguous().float()
        # ---- hr1 (k5, bias, planar out)
        h, hcat = saved["h"], saved["hcat"]
        cat_c = h.shape[-1]
        gh = self._empty(h.shape, g_out)
        # hr0's LeakyReLU + Dropout3d backward rides on the input gradient of hr1 (epilogue mask + keep factors)
        drop = self._drop_padded(saved["drop"], cat_c)
        hr0_mask = (h, 0, 0, cat_c) + ((drop,) if drop is not None else ())
        if self.zfold_active():  # adjoint of the folded forward: dy un-folded into cout*KZ channels
            kx, ky, kz = self.hr1.kernel
            cz = self.hr1.cout * kz
            cz_p = self.cp(cz)
            g3 = self._empty((B, sX, sY, nz, cz_p), g_out)
            ops.zunfold(g_out, g3, kz, self.hr1.pad[2], 0, cz_p)
            if self._hr1z_grad is None or self._hr1z_grad.device != dev:
                self._hr1z_grad = torch.empty_like(self.hr1z.weight, device=dev)
            swap = self._hr1z_t is not None and self.dt == torch.bfloat16 and cat_c == self.hr1.cin and DETERMINISTIC
            if swap:
                if self._hr1z_grad_t is None or self._hr1z_grad_t.device != dev:
                    self._hr1z_grad_t = torch.empty(self._hr1z_t.weight.shape, dtype=torch.float32, device=dev)
                self.wgrad(self._hr1z_t, g3, 0, h, 0, flat, sp, scratch, dst=self._hr1z_grad_t)
            else:
                self.wgrad(self.hr1z, h, 0, g3, 0, flat, sp, scratch, dst=self._hr1z_grad)
            self.dgrad(self.hr1z, g3, 0, gh, 0, (sX, sY, nz), mask=hr0_mask)
            self.flush_unpack()
            if swap:  # [ci][co*KZ + kz (padded)][K-1-kx][K-1-ky][0] -> hr1's [co][ci][kx][ky][kz]
                gt = self._hr1z_grad_t[:, :cz, :, :, 0].flip(2, 3).view(self.hr1.cin, self.hr1.cout, kz, kx, ky)
                sp.view(flat, self.hr1.weight).copy_(gt.permute(1, 0, 3, 4, 2))
            else:
                sp.view(flat, self.hr1.weight).copy_(
                    self._hr1z_grad.view(self.hr1.cout, kz, self.hr1.cin, kx, ky).permute(0, 2, 3, 4, 1))
        else:
            co_p = self.cp(self.hr1.cout)
            g3 = self._empty((B, sX, sY, nz, co_p), g_out)
            ops.planar_to_ndhwc(g_out, g3, 0, co_p)
            self.wgrad(self.hr1, h, 0, g3, 0, flat, sp, scratch)
            self.dgrad(self.hr1, g3, 0, gh, 0, (sX, sY, nz), mask=hr0_mask)
        ops.plane_sum(g_out, sp.view(flat, self.hr1.bias))
        ready(self.hr1.weight, self.hr1.bias)
        tr("g3", 0, g3)
        tr("gh", 0, gh)
        del g3
        # ---- hr0 (k5 + LReLU + Dropout3d mask: already applied to gh above)
        tfeat = saved.get("tfeat")  # not None: the concat is two tensors (SPLIT_CAT), and so is its gradient
        ghcat = self._empty(hcat.shape, g_out)
        if tfeat is not None:
            gtf = self._empty(tfeat.shape, g_out)
            self.wgrad(self.hr0, hcat, 0, gh, 0, flat, sp, scratch, x2=tfeat, x2_c0=nf)
            self.dgrad(self.hr0, gh, 0, ghcat, 0, (sX, sY, nz), dx2=gtf, dx2_c0=nf)
            gt_src, gt_off = gtf, 0
        else:
            self.wgrad(self.hr0, hcat, 0, gh, 0, flat, sp, scratch)
            self.dgrad(self.hr0, gh, 0, ghcat, 0, (sX, sY, nz))
            gt_src, gt_off = ghcat, nf
        ready(self.hr0.weight)
        del gh
        tr("ghcat", 0, ghcat)
        tr("gterrain", 0, gt_src[..., gt_off:gt_off + self.cp(tf)])
        # ---- terrain branch (channels nf.. of the concat)
        t0, z_nd = saved["t0"], saved["z_nd"]
        self.wgrad(self.terrain1, t0, 0, gt_src, gt_off, flat, sp, scratch)
        gt0 = self._empty(t0.shape, g_out)
        self.dgrad(self.terrain1, gt_src, gt_off, gt0, 0, (sX, sY, nz))
        del gt_src
        ops.lrelu_bwd_(gt0, 0, t0, 0, t0.shape[-1], sl)
        tr("gt0", 0, gt0)
        self.wgrad(self.terrain0, z_nd, 0, gt0, 0, flat, sp, scratch)
        ready(self.terrain1.weight, self.terrain0.weight)
        del gt0
        # ---- up-convs, last to first.  g lives in window [0, nf) of gbuf
        gbuf = ghcat
        masked = False  # gbuf already carries the LeakyReLU derivative of the up-conv output it is the gradient of
        for u in reversed(range(len(self.ups))):
            site, (inp, outp) = self.ups[u], saved["up_io"][u]
            if not masked:
                ops.lrelu_bwd_(gbuf, 0, outp, 0, nf, sl)
            masked = False
            tr("gup_out", u, gbuf[..., :nf])
            if self.subpixel_active(u) and SUBPIXEL_WGRAD and self.dt == torch.bfloat16:
                self.up_wgrad(u, inp, gbuf, flat, sp)  # (fp32: the direct 27-tap gradient - its tile kernel has no lattice form)
            else:
                self.wgrad(site, inp, 0, gbuf, 0, flat, sp, scratch)
            gin = self._empty(inp.shape, g_out)
            if self.subpixel_active(u):
                # the input of up-conv u is the LeakyReLU output of up-conv u - 1: its derivative rides on the epilogue
                # of the last parity launch instead of a pass of its own
                masked = u > 0
                self.up_dgrad(u, gbuf, gin, mask=(inp, 0, 0, nf) if masked else None)
            else:
                fine = self._empty((B, inp.shape[1] * 2, inp.shape[2] * 2, nz, nf), g_out)
                self.dgrad(site, gbuf, 0, fine, 0, tuple(inp.shape[1:4]))
                ops.upsample2_bwd(fine, gin)
                del fine
            ready(site.weight)
            tr("gup_in", u, gin)
            gbuf = gin
        if not self.ups:
            gs = self._empty((B, X, Y, nz, nf), g_out)
            ops.chan_axpby(gs, 0, ghcat, 0, nf)
            gbuf = gs
        del ghcat
        gs = gbuf  # grad of s = f + lr_conv(t_last)
        # ---- lr_conv (its input gradient is the gradient of the last RRDB's output: written straight into channels
        # [0, nf) of the dense gradient buffer when the trunk's chain starts there, see `pingpong` below)
        bufs = saved["bufs"]
        bi = len(bufs)
        dense = bufs[0].shape[-1] if bufs else nf
        gd = self._empty((B, X, Y, nz, dense), g_out) if bufs else None
        inplace = (GD_INPLACE and gd is not None and self.dt == torch.bfloat16 and self.use_tile
                   and ops.conv1x1_covers(nf, dense, True))
        pingpong = bool(GD_PINGPONG and inplace and WGRAD_STREAM < 2 and DETERMINISTIC and bufs)
        self.wgrad(self.lr_conv, saved["t_last"], 0, gs, 0, flat, sp, scratch)
        g = gd if pingpong else self._empty((B, X, Y, nz, nf), g_out)
        self.dgrad(self.lr_conv, gs, 0, g, 0, (X, Y, nz))
        ready(self.lr_conv.weight)
        tr("gs", 0, gs)
        tr("g_lr", 0, g[..., :nf])
        # ---- trunk
        # Filter gradients on a second stream (WSR_WGRAD_STREAM = ring size): they read the saved dense buffer and the
        # block's output gradients and feed nothing in the input-gradient chain.  The running gradient then MOVES through
        # a ring of dense gradient buffers - block j reads its output gradient in ring[slot][..., :nf] and writes its
        # input gradients (all `dense` channels) to ring[slot + 1] - so that the side stream may still read a buffer
        # while the chain is one or two blocks further; fences: one event per block and stream.
        side = None
        if WGRAD_STREAM >= 2 and inplace and DETERMINISTIC and bufs:
            side = self.side_stream(dev)
            main = torch.cuda.current_stream(dev)
            ring = [gd] + [self._empty(gd.shape, g_out) for _ in range(WGRAD_STREAM - 1)]
            free_ev: List[Optional[torch.cuda.Event]] = [None] * len(ring)
            slot = 0
        if pingpong:
            # Two dense gradient buffers alternate per RRDB.  `cur[..., :nf]` holds g_r, the gradient of RRDB r's output,
            # UNSCALED and untouched while the chain runs in `oth`: the first LFF input gradient of the RRDB reads it
            # there (alpha = rdb_scale * rr_scale, identity shortcut weighted rr_scale - the RRDB's residual scaling rides
            # on the launch), the last one adds it as a second residual, so that after the last block's window 0
            # oth[..., :nf] = g_r + chain gradient = g_{r-1}.  (Was: gd[:nf] = rr_scale * g and g += gd[:nf], two
            # channel-window passes per RRDB, 0.5 ms per step.)
            cur, oth = gd, self._empty(gd.shape, g_out)
            for rdbs, rr_scale in zip(reversed(self.rrdbs), reversed(self.rrdb_scales)):
                nb = len(rdbs)
                for j, (convs, lff, rdb_scale) in enumerate(reversed(rdbs)):
                    bi -= 1
                    buf = bufs[bi]
                    first, lastb = j == 0, j == nb - 1
                    go = cur if first else oth               # the block's output gradient sits in go[..., :nf]
                    sc = rdb_scale * (rr_scale if first else 1.0)
                    self.wgrad(lff, buf, 0, go, 0, flat, sp, scratch, scale=sc)
                    rows = ops.chan_sum_rows(nf, go.numel() // go.shape[-1])
                    if not rows:
                        raise RuntimeError("LFF bias gradient outside the two-pass sum (set WSR_GD_PINGPONG=0)")
                    pr = self._arena_take(rows * nf, dev).view(rows, nf)
                    ops.chan_sum_partials(go, 0, nf, pr)
                    self._pending_unpack.append((pr[0].view(1, 1, nf), sp.view(flat, lff.bias).view(1, nf, 1), sc, rows, nf))
                    nc = len(convs)
                    last = nf + (nc - 1) * gc
                    self.dgrad(lff, go, 0, oth, 0, (X, Y, nz), alpha=sc, accumulate=nf, acc_src=go if first else None,
                               acc_beta=rr_scale if first else 1.0, res2=cur if lastb else None,
                               mask=(buf, last, last, last + gc) if nc else None)
                    if STACK_DGRAD and self.dense_stackable(convs):
                        self.dgrad_dense(convs, oth, buf, (X, Y, nz))
                    else:
                        for i in reversed(range(nc)):
                            off = nf + i * gc
                            m = (buf, off - gc, off - gc, off) if i > 0 else None
                            self.dgrad(convs[i], oth, off, oth, 0, (X, Y, nz), accumulate=True, mask=m)
                    self.wgrad_dense(convs, buf, oth, flat, sp, scratch)
                    ready(lff.weight, lff.bias, *[c.weight for c in convs])
                cur, oth = oth, cur
            g = cur  # (channels [0, nf) of a dense buffer: the feature conv's filter gradient reads that window)
        for rdbs, rr_scale in zip(reversed(self.rrdbs), reversed(self.rrdb_scales)):
            if pingpong:
                break
            g_skip = g  # d(out)/d(x_rr) through the RRDB shortcut
            if side is not None:
                ops.chan_axpby(ring[slot], 0, g, 0, nf, alpha=rr_scale)
                for convs, lff, rdb_scale in reversed(rdbs):
                    bi -= 1
                    buf = bufs[bi]
                    src, nxt = ring[slot], (slot + 1) % len(ring)
                    dst = ring[nxt]
                    nc = len(convs)
                    last = nf + (nc - 1) * gc
                    ev = self._event()
                    ev.record(main)
                    side.wait_event(ev)  # the block's output gradient (src[..., :nf]) is complete
                    with torch.cuda.stream(side):
                        self.wgrad(lff, buf, 0, src, 0, flat, sp, scratch, scale=rdb_scale)
                        rows = ops.chan_sum_rows(nf, src.numel() // src.shape[-1])
                        if not rows:
                            raise RuntimeError("LFF bias gradient outside the two-pass sum (set WSR_WGRAD_STREAM=0)")
                        pr = self._arena_take(rows * nf, dev).view(rows, nf)
                        ops.chan_sum_partials(src, 0, nf, pr)
                        self._pending_unpack.append((pr[0].view(1, 1, nf), sp.view(flat, lff.bias).view(1, nf, 1),
                                                     rdb_scale, rows, nf))
                        fe = self._event()
                        fe.record(side)  # ... and the side stream's readers of `src` (this and, earlier in stream order,
                        free_ev[slot] = fe  # the previous block's stacked filter gradient) are done
                    if free_ev[nxt] is not None:  # `dst` is about to be overwritten: its side-stream readers first
                        main.wait_event(free_ev[nxt])
                        free_ev[nxt] = None
                    self.dgrad(lff, src, 0, dst, 0, (X, Y, nz), alpha=rdb_scale, accumulate=nf, acc_src=src,
                               mask=(buf, last, last, last + gc) if nc else None)
                    if STACK_DGRAD and self.dense_stackable(convs):
                        self.dgrad_dense(convs, dst, buf, (X, Y, nz))
                    else:
                        for i in reversed(range(nc)):
                            off = nf + i * gc
                            m = (buf, off - gc, off - gc, off) if i > 0 else None
                            self.dgrad(convs[i], dst, off, dst, 0, (X, Y, nz), accumulate=True, mask=m)
                    ev = self._event()
                    ev.record(main)
                    side.wait_event(ev)  # all growth-channel gradients of the block are final
                    with torch.cuda.stream(side):
                        self.wgrad_dense(convs, buf, dst, flat, sp, scratch)
                        ready(lff.weight, lff.bias, *[c.weight for c in convs])
                    slot = nxt
                ops.chan_axpby(g, 0, ring[slot], 0, nf, alpha=1.0, beta=1.0)  # g (= g_skip) += chain gradient
                continue
            if inplace:
                # the gradient of the current block's output lives in gd[..., :nf]: the LFF input gradient is
                # taken in place (1x1x1: voxel-local) with the block's identity shortcut as a residual on those
                # channels, and window 0 of the dense input gradient accumulates onto it
                go, go_c = gd, nf
                ops.chan_axpby(gd, 0, g, 0, nf, alpha=rr_scale)
            else:
                go, go_c = self._empty(g.shape, g_out), None
                ops.chan_axpby(go, 0, g, 0, nf, alpha=rr_scale)
            for convs, lff, rdb_scale in reversed(rdbs):
                bi -= 1
                buf = bufs[bi]
                tr("rdb_go", bi, go[..., :nf])
                # LFF (1x1x1, bias): out = rdb_scale * (LFF(buf) + b) + x
                self.wgrad(lff, buf, 0, go, 0, flat, sp, scratch, scale=rdb_scale)
                gb = sp.view(flat, lff.bias)
                rows = ops.chan_sum_rows(nf, go.numel() // go.shape[-1]) if DETERMINISTIC else 0
                if rows:
                    # first pass now (go is overwritten below), the sum over the partial rows rides on the ordered
                    # reduce of the filter gradients: a job with one output row and `rows` split copies
                    pr = self._arena_take(rows * nf, dev).view(rows, nf)
                    ops.chan_sum_partials(go, 0, nf, pr)
                    self._pending_unpack.append((pr[0].view(1, 1, nf), gb.view(1, nf, 1), rdb_scale, rows, nf))
                elif not ops.chan_sum(go, 0, nf, gb, scale=rdb_scale):
                    gb.copy_(go[..., :nf].float().sum(dim=(0, 1, 2, 3)) * rdb_scale)
                # gd[off_i : off_i+gc] is the gradient w.r.t. the LeakyReLU output of growth conv i; it is
                # complete once the LFF and the later convs have added into it, so the kernel that makes the
                # last contribution applies the LeakyReLU derivative in its epilogue (window i = nconv-1: the
                # LFF's input gradient; window i-1: the input gradient of conv i)
                nc = len(convs)
                last = nf + (nc - 1) * gc
                self.dgrad(lff, go, 0, gd, 0, (X, Y, nz), alpha=rdb_scale, accumulate=go_c if inplace else False,
                           mask=(buf, last, last, last + gc) if nc else None)
                if STACK_DGRAD and self.dense_stackable(convs):
                    self.dgrad_dense(convs, gd, buf, (X, Y, nz))
                else:
                    for i in reversed(range(nc)):
                        off = nf + i * gc
                        m = (buf, off - gc, off - gc, off) if i > 0 else None
                        self.dgrad(convs[i], gd, off, gd, 0, (X, Y, nz), accumulate=True, mask=m)
                # all growth-channel gradients of the block are final now: one stacked wgrad
                self.wgrad_dense(convs, buf, gd, flat, sp, scratch)
                if inplace:
                    tr("rdb_gd", bi, gd)
                if not inplace:
                    ops.chan_axpby(go, 0, gd, 0, nf, alpha=1.0, beta=1.0)  # + grad through the dense input
                ready(lff.weight, lff.bias, *[c.weight for c in convs])
            if inplace:
                ops.chan_axpby(g, 0, gd, 0, nf, alpha=1.0, beta=1.0)  # g (= g_skip) += chain gradient; g is ours
            else:
                ops.chan_axpby(go, 0, g_skip, 0, nf, alpha=1.0, beta=1.0)
                g = go
        if side is not None:  # join: everything below (and the optimizer) sees the trunk's filter gradients
            ev = self._event()
            ev.record(side)
            main.wait_event(ev)
        # ---- feature conv: total grad of f = trunk path + skip path
        ops.chan_axpby(g, 0, gs, 0, nf, alpha=1.0, beta=1.0)
        self.wgrad(self.feature, saved["x_nd"], 0, g, 0, flat, sp, scratch)
        ready(self.feature.weight)
        self.end_backward()
        if self.grad_done_hook is not None:
            self.grad_done_hook()
        return flat

    def stacked_fwd_specs(self):
        key = (self.use_tile, STACK_FWD, FWD_REGROUP)
        if self._stack_fwd_specs is None or self._stack_fwd_specs[0] != key:
            specs = [sp for rdbs in self.rrdbs for convs, _, _ in rdbs if STACK_FWD and self.dense_stackable(convs)
                     for sp in self.dense_fwd_specs(convs)]
            self._stack_fwd_specs = (key, specs)
        return self._stack_fwd_specs[1]

    def stacked_dgrad_specs(self):
        key = (self.use_tile, STACK_DGRAD)
        if self._stack_specs is None or self._stack_specs[0] != key:
            specs = [sp for rdbs in self.rrdbs for convs, _, _ in rdbs if STACK_DGRAD and self.dense_stackable(convs)
                     for sp in self.dense_dgrad_specs(convs)]
            self._stack_specs = (key, specs)
        return self._stack_specs[1]

    def _drop_padded(self, drop: Optional[Tensor], c: int) -> Optional[Tensor]:
        if drop is None or drop.shape[1] == c:
            return drop
        out = torch.zeros((drop.shape[0], c), dtype=torch.float32, device=drop.device)
        out[:, :drop.shape[1]] = drop
        return out


class _GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Z, prog: GeneratorProgram, training: bool, save: bool, drop_scale, *params):
        out, saved = prog.forward(x, Z, training, save, drop_scale)
        ctx.prog, ctx.saved = prog, saved
        return out

    @staticmethod
    def backward(ctx, g_out):
        prog, saved = ctx.prog, ctx.saved
        if saved is None:
            raise RuntimeError("Generator_3D backward without saved activations")
        flat = prog.backward(saved, g_out)
        ctx.saved = None
        grads = tuple(prog.space.view(flat, p) if p.requires_grad else None for p in prog.param_list)
        return (None, None, None, None, None, None) + grads


def run_generator(prog: GeneratorProgram, x: Tensor, Z: Tensor, training: bool,
                  drop_scale: Optional[Tensor] = None) -> Tensor:
    """``drop_scale`` (B, nf+tf) overrides the Dropout3d RNG (tests share one mask with the oracle)."""
    if drop_scale is None and training and prog.dropout_p > 0:
        keep = 1.0 - prog.dropout_p
        drop_scale = torch.bernoulli(torch.full((x.shape[0], prog.hr0.cout), keep, device=x.device)) / keep
    # autograd disables grad mode inside Function.forward, so decide here whether to keep activations
    save = torch.is_grad_enabled() and any(p.requires_grad for p in prog.param_list)
    return _GeneratorFn.apply(x, Z, prog, training, save, drop_scale, *prog.param_list)


# =============================================================================
# Discriminator feature pyramid
# =============================================================================
@dataclass
class DLayer:
    conv: ConvSite
    bn: Optional[nn.Module]  # nn.BatchNorm3d, or nn.InstanceNorm3d (normalization_type "instance": no parameters)
    act: bool

    @property
    def inorm(self) -> bool:
        return isinstance(self.bn, nn.InstanceNorm3d)


class DiscriminatorProgram(ProgramBase):
    """conv (+BatchNorm3d) + LeakyReLU pyramid of ``Discriminator_3D.features``
    (reference Discriminator_3D.py:66-169, torch_blocks.py:372-521)."""

    #: set by dist.py: all-reduce (sum) of a small fp32 tensor across the DP group, or None (backward sums)
    stat_allreduce: Optional[Callable[[Tensor], None]] = None
    #: set by dist.py: all-gather of a small fp32 vector -> (world, len) (forward statistics), or None
    stat_allgather: Optional[Callable[[Tensor], Tensor]] = None
    #: ranks in that group (equal shards per rank)
    stat_world: int = 1

    def __init__(self, layers: Sequence[DLayer], slope: float, dt: torch.dtype):
        super().__init__(dt)
        self.layers = list(layers)
        self.slope = slope
        order: List[nn.Parameter] = []
        for l in reversed(self.layers):
            if l.bn is not None and not l.inorm:
                order += [l.bn.weight, l.bn.bias]
            order.append(l.conv.weight)
        self._unit_affine: Dict[tuple, Tuple[Tensor, Tensor]] = {}
        self.space = GradSpace(order)
        self.param_list = order
        self._scratch_elems = self.wgrad_scratch_elems([l.conv for l in self.layers], self.e)
        self.all_sites = [l.conv for l in self.layers]
        self._scratch_elems_total = sum(s.cout * s.taps * self.cp(s.cin) for s in self.all_sites)
        for l in self.layers:
            if l.bn is not None and (l.conv.cout > 256 or 256 % l.conv.cout or l.conv.cout % self.e):
                raise ValueError(f"BatchNorm3d kernels need a channel count dividing 256 and a multiple of {self.e} "
                                 f"for {dt}; got {l.conv.cout}")
        # Input gradient of the down-sampling convs (reference torch_blocks.py:372-521: kernel (4,4,3), stride
        # (2,2,1|2), padding 1) in parity form: input voxel (2m+a, 2n+b, s*l+c) only meets the taps of matching
        # parity, so dx is written lattice by lattice by 2x2xKZ' forward convs over dy on the tile kernels (one launch
        # per z class, the four (a, b) parities side by side) instead of the generic gather kernel.  The parity
        # filters are tap selections of the master filter (wsr_strided_parity_filters), twins like the generator's.
        self.dparity: Dict[int, list] = {}
        self._dparity_stamp: Dict[int, object] = {}
        self._dparity_grads: Dict[tuple, list] = {}  # class gradients of the parity form of the filter gradients
        self._dparity_bad: set = set()               # (layer, shape) whose parity-form filter gradient has no tile plan
        for li, l in enumerate(self.layers):
            s = l.conv
            if li > 0 and s.kernel == (4, 4, 3) and s.pad == (1, 1, 1) and s.stride[:2] == (2, 2) and s.stride[2] in (1, 2) \
                    and s.cout % 16 == 0 and s.cin % 8 == 0:
                sz = s.stride[2]
                groups = []
                for zc in range(sz):
                    kzp = 3 if sz == 1 else (1 if zc == 0 else 2)
                    wp = torch.empty((4, s.cin, s.cout, 2, 2, kzp), dtype=torch.float32, device=s.weight.device)
                    par = [ConvSite(f"{s.name}.dparity{zc}.{ph}", wp[ph], None, (2, 2, kzp), (1, 1, 1),
                                    (1 - (ph >> 1), 1 - (ph & 1), 1 if sz == 1 else 0), fwd_only=True) for ph in range(4)]
                    groups.append([zc, wp, par])
                self.dparity[li] = groups

    def _unit(self, C_: int, dev) -> Tuple[Tensor, Tensor]:
        """(ones, zeros) of C_ floats: the affine of a normalisation layer that has none"""
        key = (C_, str(dev))
        if key not in self._unit_affine:
            self._unit_affine[key] = (torch.ones(C_, dtype=torch.float32, device=dev),
                                      torch.zeros(C_, dtype=torch.float32, device=dev))
        return self._unit_affine[key]

    def strided_dgrad_active(self, li: int) -> bool:
        return STRIDED_DGRAD and li in self.dparity and self.use_tile and self.dt == torch.bfloat16

    def strided_wgrad_active(self, li: int) -> bool:
        return STRIDED_WGRAD and li in self.dparity and self.use_tile and self.dt == torch.bfloat16 and DETERMINISTIC

    def strided_wgrad(self, li: int, inp: Tensor, gy: Tensor, flat: Tensor, space: "GradSpace") -> bool:
        """filter gradient of down-sampling conv ``li`` (reference torch_blocks.py:372-521: kernel (4,4,3), stride
        (2,2,1|2), padding 1) in parity form: tap (2i + a, 2j + b, .) only meets input voxels of one parity, so the
        taps of class (a, b[, z class]) are a stride-1 2x2xKZ' filter gradient over a sub-lattice of ``inp`` - on the
        LDS-tile kernel (x and dy fetched once per tile) instead of the generic per-tap kernel (once per tap: 10-23x
        the algorithmic traffic).  The class gradients land in twins of the master gradient and are moved to their
        taps by ``wsr_strided_parity_unfold``."""
        s = self.layers[li].conv
        sz = s.stride[2]
        B, oxyz = gy.shape[0], tuple(gy.shape[1:4])
        cin_p = self.cp(s.cin)
        key = ("dwpar", li)
        tw = self._dparity_grads.get(key)
        if tw is None or tw[0][0].device != inp.device:
            tw = [torch.empty((4, s.cout, s.cin, 2, 2, 3 if sz == 1 else (1 if zc == 0 else 2)), dtype=torch.float32,
                              device=inp.device) for zc in range(sz)]
            self._dparity_grads[key] = tw
        runs, jobs = [], []
        bad_key = (li, B) + oxyz
        if bad_key in self._dparity_bad:
            return False
        try:  # (plan queries only: nothing is taken from the arena or queued before all of them have answered)
            for zc in range(sz):
                kzp = 3 if sz == 1 else (1 if zc == 0 else 2)
                pz, mz, oz = (1, 1, 0) if sz == 1 else ((0, 2, 0) if zc == 0 else (1, 2, 1))
                for ph in range(4):
                    a_, b_ = ph >> 1, ph & 1
                    g = ConvGeom(s.cin, s.cout, (2, 2, kzp), (1, 1, 1), (1 - a_, 1 - b_, pz))
                    d = ops.make_desc(g, self.dt, B, oxyz, inp.shape[-1], 0, gy.shape[-1], 0, cin=cin_p,
                                      lat=(1 - a_, 1 - b_, 0, mz, oz, True))
                    jobs.append((zc, ph, g, d, self._wgrad_nparts(("wstr", s.name, zc, ph, B) + oxyz, d)))
        except RuntimeError:
            # the tile filter-gradient kernel declines this shape in parity form: remember it and let the caller run
            # the generic per-tap gradient (as strided_dgrad falls back to the gather kernel)
            self._dparity_bad.add(bad_key)
            return False
        # (the launches are deferred: no recycling flush between the slices)
        self._arena_reserve(sum((n * s.cout * g.taps * cin_p + 63) // 64 * 64 for _, _, g, _, n in jobs), inp.device)
        for zc, ph, g, d, n in jobs:
            parts = self._arena_take(n * s.cout * g.taps * cin_p, inp.device).view(n, s.cout, g.taps, cin_p)
            runs.append(lambda d=d, parts=parts, n=n: ops.conv_wgrad_parts(d, inp, gy, parts, n))
            self._pending_unpack.append((parts[0], tw[zc][ph], 1.0, n, parts[0].numel()))

        def run():
            for r in runs:
                r()

        if self.launch_probe is not None:
            self.launch_probe("wgrad:" + s.name, run)
        else:
            run()
        self.flush_unpack()
        dst = space.view(flat, s.weight)
        for zc in range(sz):
            ops.strided_parity_unfold(tw[zc], dst, sz, zc)
        return True

    def conv_sites(self) -> Sequence[ConvSite]:
        sites = list(self.all_sites)
        for li, groups in self.dparity.items():
            if self.strided_dgrad_active(li):
                for _, _, par in groups:
                    sites += par
        return sites

    def refresh_filters(self, backward: bool) -> None:
        if backward:
            for li, groups in self.dparity.items():
                if not self.strided_dgrad_active(li):
                    continue
                w = self.layers[li].conv.weight
                fresh = False
                for g in groups:
                    zc, wp, par = g
                    if wp.device != w.device:
                        wp = g[1] = torch.empty_like(wp, device=w.device)
                        for ph, s in enumerate(par):
                            s.weight = wp[ph]
                        fresh = True
                    fresh = self._seed_parity_frags(par, w.device) or fresh
                stamp = (w._version, w.data_ptr(), self.filters._gen)
                if fresh or stamp != self._dparity_stamp.get(li):
                    for zc, wp, par in groups:
                        ops.strided_parity_filters(w.detach().contiguous(), wp, self.layers[li].conv.stride[2], zc)
                    self.filters.touch()
                    self._dparity_stamp[li] = stamp
        super().refresh_filters(backward)

    def strided_dgrad(self, li: int, gy: Tensor, gin: Tensor, mask=None) -> None:
        """gin (B, X, Y, Z, cin) = input gradient of down-sampling conv ``li`` from gy (B, X/2, Y/2, Z/s, cout).
        ``mask`` = (y, y_off, c0, c1, slope): the LeakyReLU derivative of the layer below (whose output ``y`` is this
        conv's input) rides on the epilogues - every voxel of gin is written by exactly one parity launch."""
        s = self.layers[li].conv
        sz = s.stride[2]
        B, oxyz = gy.shape[0], tuple(gy.shape[1:4])

        def run():
            for zc, wp, par in self.dparity[li]:
                frs, batched = self._parity_frags(par)
                g = ConvGeom(s.cout, s.cin, par[0].kernel, (1, 1, 1), par[0].pad)
                if batched:
                    d = ops.make_desc(g, self.dt, B, oxyz, gy.shape[-1], 0, gin.shape[-1], 0, cin=self.cp(s.cout),
                                      cout=self.cp(s.cin), lat=(0, 0, 4, sz, zc))
                    if ops.conv_fwd_tile(d, gy, frs[0], gin, mask=mask):
                        continue
                for ph, ps in enumerate(par):
                    d = ops.make_desc(ConvGeom(s.cout, s.cin, ps.kernel, (1, 1, 1), ps.pad), self.dt, B, oxyz,
                                      gy.shape[-1], 0, gin.shape[-1], 0, cin=self.cp(s.cout), cout=self.cp(s.cin),
                                      lat=(ph >> 1, ph & 1, 0, sz, zc))
                    if not ops.conv_fwd_tile(d, gy, frs[ph], gin, mask=mask):
                        raise RuntimeError("strided input gradient outside the tile kernels (set WSR_STRIDED_DGRAD=0)")

        if self.launch_probe is not None:
            self.launch_probe("dgrad:" + s.name, run)
        else:
            run()

    def forward(self, x, training: bool, save: bool):
        """x (B, C, X, Y, Z) planar fp32 -> NDHWC feature tensor (+ saved state).

        ``x`` may be a tuple of G equally shaped inputs - the reference's D(real) and D(fake) of one iteration
        (wind_field_GAN_3D.py:247-304).  The convolutions then run ONCE on the G*B samples (the deep layers are
        latency-bound: a second sample costs them next to nothing) while BatchNorm keeps the reference's per-call
        semantics: in training mode every group of B samples gets its own batch statistics and its own running-stat
        update, in call order."""
        sl = self.slope
        xs = tuple(x) if isinstance(x, (tuple, list)) else (x,)
        G, Bg = len(xs), xs[0].shape[0]
        B = G * Bg
        ops._need_cuda(*xs, self.layers[0].conv.weight)  # inputs and parameters on the current device
        self.refresh_filters(backward=save)
        c0 = self.cp(self.layers[0].conv.cin)
        h = self._empty((B,) + tuple(xs[0].shape[2:]) + (c0,), xs[0])
        for gi, xg in enumerate(xs):
            if xg.shape != xs[0].shape:
                raise ValueError("grouped discriminator inputs must have one shape")
            ops.planar_to_ndhwc(xg.contiguous().float(), h[gi * Bg:(gi + 1) * Bg], 0, c0)
        x = xs[0]
        recs = []
        nbt: Dict[int, list] = {}  # num_batches_tracked counters to advance, by identity: [tensor, calls]
        eval_consts: Dict[int, tuple] = {}
        if not training:
            # eval-mode BatchNorm constants of ALL layers in two launches (was an add + an rsqrt per layer)
            bns = [l.bn for l in self.layers if l.bn is not None and not l.inorm]
            if bns:
                with torch.no_grad():
                    inv = torch._foreach_add([b.running_var.detach().float() for b in bns], [b.eps for b in bns]) \
                        if len({b.eps for b in bns}) > 1 else torch._foreach_add([b.running_var.detach().float() for b in bns], bns[0].eps)
                    torch._foreach_rsqrt_(inv)
                eval_consts = {id(b): (b.running_mean.detach().float(), v) for b, v in zip(bns, inv)}
        for l in self.layers:
            s = l.conv
            g = ConvGeom(s.cin, s.cout, s.kernel, s.stride, s.pad)
            oxyz = g.out_extent(*h.shape[1:4])
            cp_out = self.cp(s.cout)
            y = self._empty((B,) + oxyz + (cp_out,), x, zero=cp_out != s.cout)
            if l.bn is None:
                self.conv(s, h, 0, y, 0, act=l.act, slope=sl)
                recs.append(dict(inp=h, y=None, a=y, mean=None, invstd=None))
                h = y
                continue
            self.conv(s, h, 0, y, 0)
            bn = l.bn
            C_ = s.cout
            a = self._empty(y.shape, x)
            if l.inorm:
                # nn.InstanceNorm3d (reference torch_blocks.py:26-30; defaults: no affine, no running statistics): every
                # SAMPLE is normalised with its own per-channel statistics, in training and in eval mode alike - the
                # BatchNorm kernels on one-sample slices with unit scale, nothing to synchronise between ranks
                n = y[:1].numel() // y.shape[-1]
                one, zero = self._unit(C_, x.device)
                st = torch.empty((B, 4 * C_), dtype=torch.float32, device=x.device)
                work = torch.empty((B, 2 * C_), dtype=torch.float32, device=x.device)
                means, invstds = [], []
                for b in range(B):
                    yb = y[b:b + 1]
                    ops.bn_stats(yb, st[b, :2 * C_])
                    ops.bn_mean(st[b, :2 * C_], work[b, :C_], float(n), None)
                    ops.bn_stats(yb, st[b, 2 * C_:4 * C_], shift=work[b, :C_])
                    ops.bn_finalize(st[b, 2 * C_:4 * C_], work[b, :C_], work[b, C_:], float(n), bn.eps, 0.0, None, None, None)
                    ops.bn_apply_lrelu(yb, a[b:b + 1], work[b, :C_], work[b, C_:], one, zero, l.act, sl)
                    means.append(work[b, :C_])
                    invstds.append(work[b, C_:])
                recs.append(dict(inp=h, y=y, a=a, mean=means, invstd=invstds, count=[float(n)] * B, inorm=True))
                h = a
                continue
            if not training:
                mean, invstd = eval_consts[id(bn)]
                ops.bn_apply_lrelu(y, a, mean, invstd, bn.weight.detach(), bn.bias.detach(), l.act, sl)
                recs.append(dict(inp=h, y=y, a=a, mean=[mean], invstd=[invstd], count=[float(y.numel() // y.shape[-1])],
                                 training=False, groups=1))
                h = a
                continue
            means, invstds, counts = [], [], []
            if (self.stat_allgather is None) != (self.stat_allreduce is None):
                raise RuntimeError("SyncBN needs both statistics hooks (stat_allgather for the forward pass, "
                                   "stat_allreduce for the backward sums): set them with dist.attach()")
            n = y[:Bg].numel() // y.shape[-1]
            count = float(n)
            # pass 1: sum x -> mean ; pass 2: sum (x - mean)^2 -> variance without cancellation.
            # The per-channel steps in between are two tiny fused kernels (was ~14 torch ops per layer).
            # Every group of the pass (D(real), D(fake)) has its own statistics; they share ONE work buffer so that
            # a data-parallel run synchronises all groups of a layer with one collective.
            st = torch.empty((G, 4 * C_), dtype=torch.float32, device=x.device)  # (bn_stats overwrites; rows 16-B aligned)
            work = torch.empty((G, 2 * C_), dtype=torch.float32, device=x.device)
            track = bn.track_running_stats
            if (FUSED_BN_STATS and self.stat_allgather is None and (not track or bn.momentum is not None) and y.is_contiguous()
                    and y.shape[0] == G * Bg
                    and ops.bn_train_stats(y, G, work, bn.eps, float(bn.momentum) if track else 0.0,
                                           bn.running_mean if track else None, bn.running_var if track else None)):
                # all groups of the layer in four launches (both passes, the per-channel tails and the running-statistics
                # updates in call order: wsr_bn_train_stats) instead of six per group
                for gi in range(G):
                    yg, ag = y[gi * Bg:(gi + 1) * Bg], a[gi * Bg:(gi + 1) * Bg]
                    mean, invstd = work[gi, :C_], work[gi, C_:]
                    if track:
                        nbt.setdefault(id(bn.num_batches_tracked), [bn.num_batches_tracked, 0])[1] += 1
                    ops.bn_apply_lrelu(yg, ag, mean, invstd, bn.weight.detach(), bn.bias.detach(), l.act, sl)
                    means.append(mean)
                    invstds.append(invstd)
                    counts.append(count)
                recs.append(dict(inp=h, y=y, a=a, mean=means, invstd=invstds, count=counts, training=True, groups=G))
                h = a
                continue
            for gi in range(G):
                yg = y[gi * Bg:(gi + 1) * Bg]
                ops.bn_stats(yg, st[gi, :2 * C_])
                ops.bn_mean(st[gi, :2 * C_], work[gi, :C_], count, None)
                ops.bn_stats(yg, st[gi, 2 * C_:4 * C_], shift=work[gi, :C_])
            if self.stat_allgather is not None:
                # SyncBN in ONE collective per layer (all groups together): every rank contributes its local means
                # and its local centred second moments M2 = sum (x - mean_r)^2 (both passes above are local), and
                # the global statistics follow from the pairwise-combination rule (Chan et al.) for equal shards:
                #   mean = avg_r mean_r ,  M2 = sum_r M2_r + n * sum_r (mean_r - mean)^2
                # - no cancellation beyond the per-rank two-pass one.  (Was: two blocking all-reduces per group.)
                s2 = st[:, 2 * C_:4 * C_]
                send = torch.empty((G, 2 * C_), dtype=torch.float32, device=x.device)
                ops.bn_shard_stats(work, s2, count, send)               # {mean_r, M2_r}
                allr = self.stat_allgather(send)                         # (world, G, 2C)
                ops.bn_combine_shards(allr.view(-1, G, 2 * C_), count, work, s2)
                count = float(n) * self.stat_world
            for gi in range(G):
                yg, ag = y[gi * Bg:(gi + 1) * Bg], a[gi * Bg:(gi + 1) * Bg]
                mean, invstd = work[gi, :C_], work[gi, C_:]
                track = bn.track_running_stats
                mom = 0.0
                if track and bn.momentum is not None:
                    # (the counter does not enter the update: all layers' counters advance in one launch after the loop
                    # instead of one 4 us launch per layer and call)
                    nbt.setdefault(id(bn.num_batches_tracked), [bn.num_batches_tracked, 0])[1] += 1
                    mom = bn.momentum
                elif track:  # cumulative moving average: the factor IS the counter
                    with torch.no_grad():
                        bn.num_batches_tracked += 1
                    mom = 1.0 / float(bn.num_batches_tracked)
                ops.bn_finalize(st[gi, 2 * C_:4 * C_], mean, invstd, count, bn.eps, mom,
                                bn.running_mean if track else None, bn.running_var if track else None, None)
                ops.bn_apply_lrelu(yg, ag, mean, invstd, bn.weight.detach(), bn.bias.detach(), l.act, sl)
                means.append(mean)
                invstds.append(invstd)
                counts.append(count)
            recs.append(dict(inp=h, y=y, a=a, mean=means, invstd=invstds, count=counts, training=True, groups=G))
            h = a
        if nbt:
            with torch.no_grad():
                for calls in sorted({c for _, c in nbt.values()}):
                    torch._foreach_add_([t for t, c in nbt.values() if c == calls], calls)
        saved = dict(recs=recs, in_shape=tuple(x.shape), groups=G, group_batch=Bg) if save else None
        return h, saved

    def backward(self, saved: dict, g_feat: Tensor, need_dx: bool, need_dw: bool, lo: int = 0):
        """g_feat NDHWC (compute dtype) -> (dx planar fp32 | None, flat grads | None).  ``lo`` > 0: samples [0, lo) of
        the (grouped) batch take no part - no parameter gradients are wanted and their inputs need no gradient (the
        detached D(real) of a generator iteration): the pass runs on samples [lo, B) only; g_feat and dx cover those."""
        sl = self.slope
        dev = g_feat.device
        flat = self.space.new(dev) if need_dw else None
        scratch = None
        if need_dw:
            self.begin_backward(dev)
        sp = self.space
        done = 0
        Bg = saved.get("group_batch", g_feat.shape[0])
        if lo and (need_dw or lo % Bg):
            raise ValueError("a partial backward pass starts at a group boundary and has no parameter gradients")
        g = g_feat.contiguous()  # (covers samples [lo, B) only)
        premasked = False  # g already carries the LeakyReLU derivative of the layer it is the output gradient of
        dx = None
        recs = saved["recs"]
        bn_grad_jobs: Dict[int, tuple] = {}  # group -> (gradient slots, per-channel sums): BatchNorm weight / bias gradients

        def flush_bn_grads():
            with torch.no_grad():
                for gi in sorted(bn_grad_jobs):
                    dst, src = bn_grad_jobs[gi]
                    if gi == 0:
                        torch._foreach_copy_(dst, src)
                    else:
                        torch._foreach_add_(dst, src)
            bn_grad_jobs.clear()

        for li in reversed(range(len(self.layers))):
            l, r = self.layers[li], recs[li]
            s = l.conv
            C_ = s.cout
            act_o = r["a"][lo:]
            if self.trace is not None:
                self.trace.append(("g", li, g.clone()))
            if l.bn is None:
                if l.act and not premasked:
                    ops.lrelu_bwd_(g, 0, act_o, 0, g.shape[-1], sl)
                premasked = False
                gy = g
            else:
                bn = l.bn
                y_o = r["y"][lo:]
                gy = self._empty(y_o.shape, g)
                if r.get("inorm"):  # per-sample statistics, unit scale, no parameter gradients
                    one, _ = self._unit(C_, dev)
                    for b in range(g.shape[0]):
                        sums = torch.empty(2 * C_, dtype=torch.float32, device=dev)
                        ops.bn_bwd_reduce(g[b:b + 1], act_o[b:b + 1], y_o[b:b + 1], r["mean"][lo + b], r["invstd"][lo + b],
                                          l.act, sl, sums)
                        ops.bn_bwd_apply(g[b:b + 1], y_o[b:b + 1], gy[b:b + 1], r["mean"][lo + b], r["invstd"][lo + b], one,
                                         sums, 1.0 / r["count"][lo + b])
                elif r["training"]:
                    G = r["groups"]
                    g0 = lo // Bg
                    # the two per-channel sums of every group of the pass, side by side: one collective per layer
                    sums_all = torch.empty((G - g0, 2 * C_), dtype=torch.float32, device=dev)  # (overwritten)
                    for gi in range(g0, G):
                        o = gi * Bg - lo
                        gg, ag, yg = g[o:o + Bg], act_o[o:o + Bg], y_o[o:o + Bg]
                        sums = sums_all[gi - g0]
                        ops.bn_bwd_reduce(gg, ag, yg, r["mean"][gi], r["invstd"][gi], l.act, sl, sums)
                        if need_dw:  # (every group's batch statistics are a call of their own: the gradients add)
                            # (written behind the layer loop: one foreach copy for the first group of every layer, one
                            # foreach add per further group - was four 4 us launches per layer)
                            bn_dst = [sp.view(flat, bn.bias), sp.view(flat, bn.weight)]
                            bn_grad_jobs.setdefault(gi, ([], []))
                            bn_grad_jobs[gi][0].extend(bn_dst)
                            bn_grad_jobs[gi][1].extend([sums[:C_], sums[C_:]])
                    if self.stat_allreduce is not None:
                        sums_all = sums_all.clone()
                        self.stat_allreduce(sums_all)
                    for gi in range(g0, G):
                        o = gi * Bg - lo
                        gg, yg, gyg = g[o:o + Bg], y_o[o:o + Bg], gy[o:o + Bg]
                        ops.bn_bwd_apply(gg, yg, gyg, r["mean"][gi], r["invstd"][gi], bn.weight.detach(),
                                         sums_all[gi - g0], 1.0 / r["count"][gi])
                else:
                    mean, invstd = r["mean"][0], r["invstd"][0]
                    # (no parameter gradients wanted - D inside a generator iteration: the LeakyReLU derivative rides
                    # on the BatchNorm pass instead of a pass of its own)
                    fused = (l.act and not need_dw and
                             ops.bn_bwd_apply(g, y_o, gy, mean, invstd, bn.weight.detach(), None, 0.0, act_y=act_o, slope=sl))
                    if not fused:
                        if l.act:
                            ops.lrelu_bwd_(g, 0, act_o, 0, g.shape[-1], sl)
                        if need_dw:  # eval-mode BN: d beta = sum g, d gamma = sum g * xhat
                            sums = torch.empty(2 * C_, dtype=torch.float32, device=dev)
                            ops.bn_bwd_reduce(g, act_o, y_o, mean, invstd, False, sl, sums)
                            sp.view(flat, bn.bias).copy_(sums[:C_])
                            sp.view(flat, bn.weight).copy_(sums[C_:])
                        ops.bn_bwd_apply(g, y_o, gy, mean, invstd, bn.weight.detach(), None, 0.0)
            if self.trace is not None:
                self.trace.append(("gy", li, gy.clone()))
            inp = r["inp"][lo:]
            lattice_ok = (li in self.dparity and tuple(inp.shape[1:3]) == (2 * gy.shape[1], 2 * gy.shape[2])
                          and inp.shape[3] == s.stride[2] * gy.shape[3] and inp.shape[-1] == self.cp(s.cin))
            if need_dw:
                if not (lattice_ok and self.strided_wgrad_active(li) and self.strided_wgrad(li, inp, gy, flat, sp)):
                    self.wgrad(s, inp, 0, gy, 0, flat, sp, scratch)
                if self.grad_ready_hook is not None:
                    flush_bn_grads()  # (the slots may lie inside the range handed over)
                    hi = sp.offsets[id(s.weight)][0] + (s.weight.numel() + 63) // 64 * 64
                    self.grad_ready_hook(flat, done, hi, self.flush_unpack)
                    done = hi
            if li > 0:
                gin = self._empty(inp.shape, g)
                if self.strided_dgrad_active(li) and lattice_ok:
                    # the layer below has no BatchNorm (the first conv: conv + LeakyReLU): its leaky_relu_backward rides on
                    # the epilogues of this input gradient instead of a pass of its own over the largest tensor of D
                    below = self.layers[li - 1]
                    premasked = (below.bn is None and below.act and inp.shape[-1] == below.conv.cout
                                 and below.conv.cout <= 32 and FOLD_D_MASK)
                    self.strided_dgrad(li, gy, gin, mask=(recs[li - 1]["a"][lo:], 0, 0, below.conv.cout, sl)
                                       if premasked else None)
                else:
                    self.dgrad(s, gy, 0, gin, 0, tuple(inp.shape[1:4]))
                if self.trace is not None:
                    self.trace.append(("gin", li, gin.clone()))
                g = gin
            elif need_dx:
                dx = torch.empty((inp.shape[0],) + tuple(saved["in_shape"][1:]), dtype=torch.float32, device=dev)
                self.dgrad(s, gy, 0, dx, 0, tuple(inp.shape[1:4]), dx_planar=True)
        if need_dw:
            flush_bn_grads()
            self.end_backward()
        if need_dw and self.grad_done_hook is not None:
            self.grad_done_hook()
        return dx, flat


class _DiscriminatorFeaturesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, prog: DiscriminatorProgram, training: bool, save: bool, *params):
        feat, saved = prog.forward(x, training, save)
        ctx.prog, ctx.saved = prog, saved
        ctx.feat_c = prog.layers[-1].conv.cout
        # logical (B, C, X, Y, Z) fp32 view for the (tiny) classifier head
        return feat[..., :ctx.feat_c].permute(0, 4, 1, 2, 3).float()

    @staticmethod
    def backward(ctx, g_out):
        prog, saved = ctx.prog, ctx.saved
        if saved is None:
            raise RuntimeError("Discriminator_3D.features backward without saved activations")
        need_dx = ctx.needs_input_grad[0]
        need_dw = any(ctx.needs_input_grad[4:])
        c = ctx.feat_c
        cp_ = prog.cp(c)
        g = g_out.permute(0, 2, 3, 4, 1)
        if cp_ != c:
            gp = torch.zeros(g.shape[:-1] + (cp_,), dtype=prog.dt, device=g.device)
            gp[..., :c] = g
            g = gp
        else:
            g = g.to(prog.dt).contiguous()
        dx, flat = prog.backward(saved, g, need_dx, need_dw)
        ctx.saved = None
        grads = tuple(prog.space.view(flat, p) if (need_dw and ctx.needs_input_grad[4 + i]) else None
                      for i, p in enumerate(prog.param_list))
        return (dx, None, None, None) + grads


class _DiscriminatorPairFn(torch.autograd.Function):
    """features of two inputs in ONE batched pass (see DiscriminatorProgram.forward): (xa, xb) -> (2B, C, X, Y, Z)"""

    @staticmethod
    def forward(ctx, xa, xb, prog: DiscriminatorProgram, training: bool, save: bool, *params):
        feat, saved = prog.forward((xa, xb), training, save)
        ctx.prog, ctx.saved = prog, saved
        ctx.feat_c = prog.layers[-1].conv.cout
        ctx.bg = xa.shape[0]
        return feat[..., :ctx.feat_c].permute(0, 4, 1, 2, 3).float()

    @staticmethod
    def backward(ctx, g_out):
        prog, saved = ctx.prog, ctx.saved
        if saved is None:
            raise RuntimeError("Discriminator_3D.features backward without saved activations")
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_dw = any(ctx.needs_input_grad[5:])
        c = ctx.feat_c
        cp_ = prog.cp(c)
        # the first input's half of the pass is skipped when nothing wants it (generator iteration: D(real) is detached)
        lo = 0 if (need_a or need_dw) else ctx.bg
        g = g_out[lo:].permute(0, 2, 3, 4, 1)
        if cp_ != c:
            gp = torch.zeros(g.shape[:-1] + (cp_,), dtype=prog.dt, device=g.device)
            gp[..., :c] = g
            g = gp
        else:
            g = g.to(prog.dt).contiguous()
        dx, flat = prog.backward(saved, g, need_a or need_b, need_dw, lo=lo)
        ctx.saved = None
        dxa = dxb = None
        if dx is not None:
            if lo:
                dxb = dx
            else:
                dxa, dxb = (dx[:ctx.bg] if need_a else None), (dx[ctx.bg:] if need_b else None)
        grads = tuple(prog.space.view(flat, p) if (need_dw and ctx.needs_input_grad[5 + i]) else None
                      for i, p in enumerate(prog.param_list))
        return (dxa, dxb, None, None, None) + grads


def run_discriminator_features_pair(prog: DiscriminatorProgram, xa: Tensor, xb: Tensor, training: bool) -> Tensor:
    save = torch.is_grad_enabled() and (xa.requires_grad or xb.requires_grad
                                        or any(p.requires_grad for p in prog.param_list))
    return _DiscriminatorPairFn.apply(xa, xb, prog, training, save, *prog.param_list)


def run_discriminator_features(prog: DiscriminatorProgram, x: Tensor, training: bool) -> Tensor:
    save = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in prog.param_list))
    return _DiscriminatorFeaturesFn.apply(x, prog, training, save, *prog.param_list)
