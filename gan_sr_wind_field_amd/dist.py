"""Data parallelism for the GAN train step: one process per GPU over
``torch.distributed`` (backend "nccl" = RCCL over xGMI on MI355X; "gloo" for tests).

The reference has no distributed code (single ``cuda:{gpu_id}``); this is the
extension the MI355X build adds.  ``attach(gan)`` makes an N-rank step equal the
single-GPU step on the concatenated batch:

* G / D filter gradients live in the programs' flat fp32 buffers laid out in backward
  production order; as ranges become final they are all-reduced (average) in
  ``bucket_mb`` buckets, asynchronously on the collective's own stream, while the
  rest of backward keeps the compute stream busy.  xGMI is point-to-point, a ring
  all-reduce is per-link bound, so buckets are large (default 32 MB: 139 MB of G
  gradients = 5 collectives) rather than DDP's 25 MB-of-small-tensors default;
* BatchNorm3d batch statistics are synchronised (SyncBN) with ONE collective per layer and pass: the forward
  all-gathers every rank's (mean, centred second moment) and combines them exactly, the backward sum-reduces
  its two sums; the RaGAN average logits are batch-global means with a matching
  backward, the four physics-loss normalisers are max-reduced;
* parameters and BN buffers are broadcast from rank 0 at attach time; RNG streams
  for dropout / instance noise are offset per rank.

No collective is issued on the data path of a forward conv: samples are independent.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist

Tensor = torch.Tensor

#: ledgers of the attached DataParallel objects by process group (the autograd functions below only know their group)
_LEDGERS: dict = {}


def _gkey(group) -> int:
    return 0 if group is None else id(group)


def _note(kind: str, t: "Tensor", group=None) -> None:
    st = _LEDGERS.get(_gkey(group))
    if st is not None:
        st.add(kind, t)


def _timed(kind: str, t: "Tensor", group=None):
    st = _LEDGERS.get(_gkey(group))
    return st.bracket(kind, t) if st is not None else _NULL_BRACKET


def init_from_env(backend: Optional[str] = None, single_rank: bool = False) -> bool:
    """Initialise the default process group from torchrun's environment.  Returns False
    (and does nothing) for a single-process run - unless ``single_rank``: then a group of ONE rank is set up, so that
    every collective of the data-parallel step (gradient buckets, SyncBN statistics, loss scalars) goes through the
    real transport (RCCL on a one-GPU box) and can be checked / timed there."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not single_rank:
        return False
    if world <= 1:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("LOCAL_RANK", "0")
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        if world <= 1:
            # a group of one needs no TCP store: a private file keeps two such runs on one box (or one next to a real
            # torchrun job on the default port) from colliding on MASTER_PORT
            import atexit
            import shutil
            import tempfile
            d = tempfile.mkdtemp(prefix="wsr_pg_")  # a directory of our own: no name-reuse race, nothing left behind
            atexit.register(shutil.rmtree, d, True)
            dist.init_process_group(backend=backend, init_method=f"file://{os.path.join(d, 'store')}", rank=0,
                                    world_size=1)
        else:
            dist.init_process_group(backend=backend)
    return True


class _BatchMean(torch.autograd.Function):
    """mean over the global batch: forward all-reduces (sum, count); backward all-reduces
    the upstream gradient because every rank's loss depends on every rank's logits."""

    @staticmethod
    def forward(ctx, t: Tensor, group):
        # equal shards per rank: the global element count is numel * world (no device round trip)
        total = t.sum().float().reshape(1)
        _note("scalar", total, group)
        with _timed("scalar", total, group):
            dist.all_reduce(total, group=group)
        count = float(t.numel() * dist.get_world_size(group))
        ctx.group, ctx.count, ctx.shape, ctx.dtype = group, count, t.shape, t.dtype
        return (total[0] / count).to(t.dtype)

    @staticmethod
    def backward(ctx, g: Tensor):
        g = g.clone().float()
        _note("scalar", g, ctx.group)
        with _timed("scalar", g, ctx.group):
            dist.all_reduce(g, group=ctx.group)
        # the later gradient average over ranks divides by N once more, exactly as it does
        # for every other term of the per-rank mean losses
        return (g / ctx.count).to(ctx.dtype).expand(ctx.shape), None


class _BatchMean2(torch.autograd.Function):
    """(mean a, mean b) over the global batch in ONE collective per pass - the two average logits of the relativistic
    average GAN losses (reference wind_field_GAN_3D.py:360-364, 552-556) are needed together: 2 + 2 scalar
    collectives per loss become 1 + 1."""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor, group, dp=None):
        tot = torch.stack([a.sum().float(), b.sum().float()])
        _note("scalar", tot, group)
        with _timed("scalar", tot, group):
            dist.all_reduce(tot, group=group)
        world = dist.get_world_size(group)
        ctx.group, ctx.dp = group, dp
        ctx.meta = ((float(a.numel() * world), a.shape, a.dtype), (float(b.numel() * world), b.shape, b.dtype))
        return (tot[0] / ctx.meta[0][0]).to(a.dtype), (tot[1] / ctx.meta[1][0]).to(b.dtype)

    @staticmethod
    def backward(ctx, ga: Tensor, gb: Tensor):
        g = torch.stack([ga.reshape(()).float(), gb.reshape(()).float()])
        g, rider = _take_rider(ctx.dp, g)
        _note("scalar", g, ctx.group)
        with _timed("scalar", g, ctx.group):
            dist.all_reduce(g, group=ctx.group)
        _give_rider(rider, g, 2)
        (ca, sa, da), (cb, sb, db) = ctx.meta
        return (g[0] / ca).to(da).expand(sa), (g[1] / cb).to(db).expand(sb), None, None


def _take_rider(dp, g: Tensor):
    """(vector to all-reduce, rider) - a small non-differentiable vector that waits for the next backward collective of
    the loss scalars travels in the same SUM all-reduce (``DataParallel.ride``: the generator iteration's guard flags)"""
    rider = getattr(dp, "_rider", None) if dp is not None else None
    if rider is None:
        return g, None
    dp._rider = None
    return torch.cat([g, rider[0].to(g.dtype).reshape(-1)]), rider


def _give_rider(rider, g: Tensor, n: int) -> None:
    if rider is not None:
        rider[1](g[n:])


class _MeansMax(torch.autograd.Function):
    """(mean a, mean b, element-wise max m) over the global batch in ONE collective: every rank contributes the record
    [sum a, sum b, m...] to an all-gather and reduces the gathered rows itself (sums for the means, maxima for m; a NaN
    in any rank's maximum propagates, as ``torch.max`` does on one GPU).  The generator iteration needs the two RaGAN
    average logits and the eight physics-loss maxima before it can form its loss (reference
    wind_field_GAN_3D.py:360-364, 773-814): two blocking collectives become one.  Backward: one SUM all-reduce of the
    two mean gradients (plus a rider, see ``DataParallel.ride``); the maxima are not differentiated here (the fused
    content-loss path; the composed path keeps :class:`_GlobalMax`)."""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor, m: Tensor, dp):
        rec = torch.cat([a.sum().float().reshape(1), b.sum().float().reshape(1), m.detach().float().reshape(-1)])
        rows = dp.gather_small(rec, "scalar")
        tot = rows[:, :2].sum(0)
        gmax = rows[:, 2:].max(0).values.reshape(m.shape).to(m.dtype)
        ctx.dp = dp
        ctx.meta = ((float(a.numel() * dp.world), a.shape, a.dtype), (float(b.numel() * dp.world), b.shape, b.dtype))
        ctx.mark_non_differentiable(gmax)
        return (tot[0] / ctx.meta[0][0]).to(a.dtype), (tot[1] / ctx.meta[1][0]).to(b.dtype), gmax

    @staticmethod
    def backward(ctx, ga: Tensor, gb: Tensor, _gm):
        g = torch.stack([ga.reshape(()).float(), gb.reshape(()).float()])
        g, rider = _take_rider(ctx.dp, g)
        _note("scalar", g, ctx.dp.group)
        with _timed("scalar", g, ctx.dp.group):
            dist.all_reduce(g, group=ctx.dp.group)
        _give_rider(rider, g, 2)
        (ca, sa, da), (cb, sb, db) = ctx.meta
        return (g[0] / ca).to(da).expand(sa), (g[1] / cb).to(db).expand(sb), None, None


class _GlobalMax(torch.autograd.Function):
    """element-wise maximum over the ranks that stays in the autograd graph: the value is the all-reduced
    maximum, the gradient - summed over the ranks, because every rank's loss uses the global value - goes to
    the rank (and, through ``torch.max`` upstream, the element) that holds the maximum.  This is what the
    single-GPU step does on the concatenated batch (reference wind_field_GAN_3D.py:773-814 keeps the maxima
    attached).  Ties between ranks (measure zero) would each receive the full gradient."""

    @staticmethod
    def forward(ctx, t: Tensor, group):
        g = t.detach().clone()
        _note("scalar", g, group)
        with _timed("scalar", g, group):
            dist.all_reduce(g, op=dist.ReduceOp.MAX, group=group)
        ctx.group = group
        ctx.save_for_backward(t.detach() == g)
        return g

    @staticmethod
    def backward(ctx, grad: Tensor):
        (owner,) = ctx.saved_tensors
        grad = grad.clone()
        _note("scalar", grad, ctx.group)
        with _timed("scalar", grad, ctx.group):
            dist.all_reduce(grad, group=ctx.group)
        return grad * owner.to(grad.dtype), None


class CommStats:
    """Per-process communication ledger of the data-parallel step (read by ``bench.py`` into its ``comm`` object).

    ``count`` / ``bytes`` per kind of collective ("grad" = gradient buckets, "syncbn", "scalar" = RaGAN batch means,
    physics-loss maxima and the OR-ed guard flags).  With ``timing`` on, every BLOCKING collective and every
    ``DataParallel.wait()`` is bracketed by two HIP events on the compute stream: the elapsed time between them is
    what the compute stream spent behind the collective - the *exposed* communication time (an asynchronous
    gradient bucket that finished under the backward pass shows up as ~0 in ``wait``).  Events are resolved by
    ``summary()`` after the caller has synchronised; nothing here blocks the host inside the step."""

    KINDS = ("grad", "syncbn", "scalar")

    def __init__(self):
        self.timing = False
        self.reset()

    def reset(self) -> None:
        self.count = {k: 0 for k in self.KINDS}
        self.bytes = {k: 0 for k in self.KINDS}
        self._events = {"wait": [], "syncbn": [], "scalar": []}
        self.retries = 0  # generator passes discarded by the loss guards and run again (their collectives are counted)

    def add(self, kind: str, t: Tensor) -> None:
        self.count[kind] += 1
        self.bytes[kind] += t.numel() * t.element_size()

    def bracket(self, kind: str, t: Tensor):
        """context manager: HIP events around a region of the current stream (no-op unless timing a device tensor)"""
        return _Bracket(self, kind) if (self.timing and t.is_cuda) else _NULL_BRACKET

    def summary(self, steps: int) -> dict:
        """per-step figures; call after torch.cuda.synchronize()"""
        ms = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in self._events.items()}
        per = lambda x: round(x / max(1, steps), 3)  # noqa: E731
        return {
            "collectives_per_step": per(sum(self.count.values())),
            "grad_bucket_collectives_per_step": per(self.count["grad"]),
            "grad_mbytes_per_step": per(self.bytes["grad"] / 1e6),
            "syncbn_collectives_per_step": per(self.count["syncbn"]),
            "syncbn_kbytes_per_step": per(self.bytes["syncbn"] / 1e3),
            "scalar_collectives_per_step": per(self.count["scalar"]),
            "exposed_grad_wait_ms_per_step": per(ms["wait"]) if self.timing else None,
            "syncbn_blocking_ms_per_step": per(ms["syncbn"]) if self.timing else None,
            "scalar_blocking_ms_per_step": per(ms["scalar"]) if self.timing else None,
            "timed": self.timing,
            "discarded_generator_passes_per_step": per(self.retries),
        }


class _Bracket:
    def __init__(self, stats: "CommStats", kind: str):
        self.stats, self.kind = stats, kind

    def __enter__(self):
        self.e0 = torch.cuda.Event(enable_timing=True)
        self.e0.record()

    def __exit__(self, *exc):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.stats._events[self.kind].append((self.e0, e1))
        return False


class _NullBracket:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_BRACKET = _NullBracket()


class DataParallel:
    def __init__(self, group=None, bucket_mb: float = 32.0, sync_bn: bool = True, tail_mb: float = 1.0):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (use init_from_env or torchrun)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bucket_elems = max(1, int(bucket_mb * 1024 * 1024 / 4))
        self.sync_bn = sync_bn
        self._avg_native = dist.get_backend(group) == "nccl"
        self._pending: List = []      # (work, tensor) of in-flight gradient buckets
        self._open = {}               # program id -> [flat, lo, hi] of the bucket being filled
        self.n_collectives = 0        # bookkeeping for tests / DESIGN.md numbers
        #: buckets shrink towards the end of a backward pass: the final collective cannot hide under anything, so a bucket
        #: also closes once it holds at least as much as everything still to come (>= tail_mb) - 32, 32, ..., 16, 8, 4, 2, 1
        self.tail_elems = max(1, int(tail_mb * 1024 * 1024 / 4))
        self._rider = None
        self._small: List = []        # tiny gradients waiting to travel together (see _avg_param)
        self._small_back: List = []   # (flat, [tensors]) of coalesced collectives in flight
        self.stats = CommStats()
        _LEDGERS[_gkey(group)] = self.stats  # (one ledger per process group: a later DataParallel on another group keeps its own)

    # ---- collectives used inside the step ----------------------------------------
    def batch_mean(self, t: Tensor) -> Tensor:
        return _BatchMean.apply(t, self.group)

    def batch_means(self, a: Tensor, b: Tensor):
        """(global mean of a, global mean of b) with one collective per pass"""
        return _BatchMean2.apply(a, b, self.group, self)

    def means_and_max(self, a: Tensor, b: Tensor, m: Tensor):
        """(global mean of a, global mean of b, element-wise global maximum of m) with ONE collective forward and one
        backward (see :class:`_MeansMax`); ``m`` is not differentiated"""
        return _MeansMax.apply(a, b, m, self)

    def gather_small(self, rec: Tensor, kind: str = "scalar") -> Tensor:
        """(world, len(rec)) - every rank's copy of a small vector, one collective"""
        out = torch.empty((self.world,) + tuple(rec.shape), dtype=rec.dtype, device=rec.device)
        self.stats.add(kind, rec)
        with self.stats.bracket(kind, rec):
            if self._avg_native:  # nccl / RCCL
                dist.all_gather_into_tensor(out, rec.contiguous(), group=self.group)
            else:
                dist.all_gather(list(out.unbind(0)), rec.contiguous(), group=self.group)
        return out

    def ride(self, flags: Tensor, on_result) -> None:
        """``flags`` (small non-negative float vector, "set on any rank" semantics) travels in the NEXT backward
        collective of the loss scalars instead of taking one of its own: that all-reduce is a SUM, and a sum of 0 / 1
        flags is > 0 exactly when one rank set it.  ``on_result(summed)`` is called inside that backward, right behind
        the collective.  A caller whose backward pass may hold no such collective asks ``take_unridden`` afterwards."""
        assert getattr(self, "_rider", None) is None, "a rider is still waiting: its backward pass never ran a scalar collective"
        self._rider = (flags, on_result)

    def take_unridden(self):
        """the rider no backward collective picked up (``None`` when it travelled), cleared"""
        r, self._rider = getattr(self, "_rider", None), None
        return r

    def global_max(self, t: Tensor) -> Tensor:
        """maximum over the ranks; differentiable when ``t`` is (see :class:`_GlobalMax`)"""
        if t.requires_grad and torch.is_grad_enabled():
            return _GlobalMax.apply(t, self.group)
        t = t.detach().clone()
        self.stats.add("scalar", t)
        with self.stats.bracket("scalar", t):
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    def stat_allreduce(self, t: Tensor) -> None:
        self.stats.add("syncbn", t)
        with self.stats.bracket("syncbn", t):
            dist.all_reduce(t, group=self.group)

    def stat_allgather(self, t: Tensor) -> Tensor:
        """(world, len(t)) copies of a small vector from every rank - the one collective of a SyncBN forward"""
        return self.gather_small(t, "syncbn")

    def _avg_async(self, t: Tensor) -> None:
        if self._avg_native:
            work = dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        else:
            work = dist.all_reduce(t, group=self.group, async_op=True)
        self._pending.append((work, t))
        self.n_collectives += 1
        self.stats.add("grad", t)

    # ---- gradient buckets ------------------------------------------------------------
    def grad_ready(self, key, flat: Tensor, lo: int, hi: int, before_launch=None) -> None:
        """range [lo, hi) of ``flat`` is final (once ``before_launch()`` - the program's pending moves into the flat
        buffer - has run); launch a collective once a bucket is full"""
        cur = self._open.get(key)
        if cur is None or cur[0] is not flat:
            cur = [flat, lo, lo]
            self._open[key] = cur
        cur[2] = hi
        size, remaining = cur[2] - cur[1], flat.numel() - hi
        if size >= self.bucket_elems or (size >= self.tail_elems and size >= remaining):
            if before_launch is not None:
                before_launch()
            self._avg_async(flat[cur[1]:cur[2]])
            cur[1] = cur[2]

    def grad_done(self, key) -> None:
        """end of a program's backward: flush the partial bucket and wait for all of them"""
        cur = self._open.pop(key, None)
        if cur is not None and cur[2] > cur[1]:
            self._avg_async(cur[0][cur[1]:cur[2]])
        self.wait()

    def _avg_param(self, g: Tensor) -> None:
        """gradient of a parameter outside the flat buffers (the classifier head): large ones go out at once, the few tiny
        ones (two biases, a 100-element weight) wait for `wait()` and travel together - one collective instead of three"""
        if g.numel() >= 16384:
            self._avg_async(g)
        else:
            self._small.append(g)

    def _flush_small(self) -> None:
        if not self._small:
            return
        small, self._small = self._small, []
        flat = torch.cat([g.reshape(-1) for g in small])
        self._avg_async(flat)
        self._small_back.append((flat, small))

    def wait(self) -> None:
        """make the compute stream wait for every gradient bucket in flight (the host does not block on RCCL)"""
        self._flush_small()
        if not self._pending:
            return
        with self.stats.bracket("wait", self._pending[0][1]):
            for work, t in self._pending:
                work.wait()
        if not self._avg_native:
            for _, t in self._pending:
                t.div_(self.world)
        self._pending.clear()
        for flat, small in self._small_back:  # the coalesced tiny gradients return to their tensors
            off = 0
            for g in small:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        self._small_back.clear()

    # ---- wiring --------------------------------------------------------------------------
    def broadcast_module(self, module: torch.nn.Module) -> None:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=0, group=self.group)
        # writes through .data do not bump the version counters the packed-filter caches watch: a forward that
        # ran before attach() (a warm-up, a validation pass after load_model) would leave ranks != 0 with
        # stale compute copies of the pre-broadcast weights
        for m in module.modules():
            prog = getattr(m, "_program", None)
            if prog is not None and hasattr(prog, "filters"):
                prog.filters.invalidate()

    def attach(self, gan) -> "DataParallel":
        """Wire a ``wind_field_GAN_3D`` for data-parallel training."""
        gan.dp = self
        self.broadcast_module(gan.G)

        def hip_backed(module) -> bool:
            # the fused HIP programs produce gradients in flat buckets; anything else (the classifier
            # head, or CPU stand-in networks in the gloo tests) goes through per-parameter hooks
            return hasattr(module, "program") and next(module.parameters()).is_cuda

        def hook_params(module):
            for p in module.parameters():
                p.register_post_accumulate_grad_hook(lambda p_: self._avg_async(p_.grad))

        progG = gan.G.program() if hip_backed(gan.G) else None
        if progG is None:
            hook_params(gan.G)
        if progG is not None:
            progG.grad_ready_hook = lambda flat, lo, hi, pre=None: self.grad_ready("G", flat, lo, hi, pre)
            progG.grad_done_hook = lambda: self.grad_done("G")
        if getattr(gan, "D", None) is not None:
            self.broadcast_module(gan.D)
            feats = gan.D.features
            progD = feats.program() if hip_backed(feats) else None
            if progD is None:
                hook_params(feats)
            if progD is not None:
                progD.grad_ready_hook = lambda flat, lo, hi, pre=None: self.grad_ready("D", flat, lo, hi, pre)
                progD.grad_done_hook = lambda: self.grad_done("D")
                if self.sync_bn:
                    progD.stat_allreduce = self.stat_allreduce
                    progD.stat_allgather = self.stat_allgather
                    progD.stat_world = self.world
            # the classifier head is ordinary torch autograd: reduce its 4 small tensors per step
            for p in gan.D.classifier.parameters():
                p.register_post_accumulate_grad_hook(lambda p_: self._avg_param(p_.grad))
        for opt in getattr(gan, "optimizers", []):
            opt.register_step_pre_hook(lambda *_: self.wait())
        # decorrelate dropout / instance-noise streams across ranks
        torch.manual_seed(torch.initial_seed() + 7919 * self.rank)
        return self


def attach(gan, bucket_mb: float = 32.0, sync_bn: bool = True, group=None) -> DataParallel:
    return DataParallel(group, bucket_mb, sync_bn).attach(gan)
