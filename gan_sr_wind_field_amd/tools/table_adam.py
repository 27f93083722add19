"""torch.optim.Adam with a one-launch step on the HIP path.

The reference builds two ``torch.optim.Adam`` (GAN_models/wind_field_GAN_3D.py:151-162) and calls ``.step()`` once per
iteration (:460, :566).  torch's fused multi-tensor implementation packs tensor pointers into kernel arguments, so the
generator's 297 tensors take 8 launches of ~78 workgroups (0.39 ms of a 93 ms step at 0.97 GB of traffic); here the
pointers live in a DEVICE table (``wsr_adam_multi``, one workgroup per 32 768-element chunk) and the whole parameter list
is one launch.  Same hyper-parameters, same ``state`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter) and
therefore the same ``state_dict`` and checkpoints as ``torch.optim.Adam(fused=True)``, which it falls back to whenever
the fast path does not apply (CPU tensors, a missing gradient, amsgrad / maximize / capturable / differentiable).
"""
from typing import Dict, List, Tuple

import torch

from .. import hip_ops


class TableAdam(torch.optim.Adam):
    def __init__(self, params, **kw):
        kw.setdefault("fused", True)
        super().__init__(params, **kw)
        self._tables: Dict[int, torch.Tensor] = {}   # per param group: device table of (param, grad, state) chunks
        self._sig: Dict[int, tuple] = {}             # ... and the (param, grad) pointers it was built from
        self._host_step: List[int] = [-1] * len(self.param_groups)  # -1: not yet read from the state
        self._steps_dirty = False

    # ---- state bookkeeping ----------------------------------------------------------------------------------
    def _init_state(self, p: torch.Tensor) -> dict:
        st = self.state[p]
        if len(st) == 0:  # as torch.optim.Adam._init_group does for fused=True
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _sync_steps(self) -> None:
        """write the host step counts into the per-parameter ``step`` tensors (before anything reads the state)"""
        if not self._steps_dirty:
            return
        for gi, group in enumerate(self.param_groups):
            steps = [self.state[p]["step"] for p in group["params"] if "step" in self.state.get(p, {})]
            if steps and self._host_step[gi] >= 0:
                torch._foreach_zero_(steps)
                torch._foreach_add_(steps, float(self._host_step[gi]))
        self._steps_dirty = False

    def _fast_ok(self, group: dict) -> bool:
        if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable") \
                or group.get("decoupled_weight_decay"):  # (the kernel implements Adam's L2 form of weight decay only)
            return False
        if not isinstance(group["lr"], float) and not isinstance(group["lr"], int):
            return False
        for p in group["params"]:
            g = p.grad
            if g is None or g.is_sparse or not p.is_cuda or p.dtype != torch.float32 or g.dtype != torch.float32 \
                    or not p.is_contiguous() or not g.is_contiguous() or g.device != p.device:
                return False
        return len(group["params"]) > 0

    # ---- torch.optim.Optimizer interface ----------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        # (host cost matters at the small presets, where ~1 600 launches per 30 ms step leave the host no slack: one pass
        # over the pointers decides whether last step's table still describes this one - then nothing else is looked at)
        # ... the pointers AND what _fast_ok decides from the group's options: a learning rate that became a tensor, a
        # flag switched on after construction must not ride on a table built before
        sigs = [(type(g["lr"]) in (float, int), bool(g.get("amsgrad") or g.get("maximize") or g.get("capturable")
                                                      or g.get("differentiable") or g.get("decoupled_weight_decay")))
                + tuple((p.data_ptr(), p.grad.data_ptr()) if p.grad is not None else None for p in g["params"])
                for g in self.param_groups]
        hit = closure is None and all(self._sig.get(gi) == sig for gi, sig in enumerate(sigs))
        fast = hit or (closure is None and all(self._fast_ok(g) for g in self.param_groups))
        if fast:
            for gi, group in enumerate(self.param_groups):
                if self._host_step[gi] < 0:  # first fast step (or after a fallback / load_state_dict): one read of the state
                    steps = torch.stack([self._init_state(p)["step"].float() for p in group["params"]])
                    lo, hi = (float(v) for v in torch.stack([steps.min(), steps.max()]).tolist())
                    if lo != hi:  # parameters of one group at different step counts: only torch's per-tensor form is right
                        fast = False
                        break
                    self._host_step[gi] = int(round(hi))
        if not fast:
            self._sync_steps()
            out = super().step(closure)
            self._host_step = [-1] * len(self.param_groups)
            self._sig.clear()
            return out
        for gi, group in enumerate(self.param_groups):
            self._host_step[gi] += 1
            if self._sig.get(gi) != sigs[gi]:
                quads: List[Tuple[torch.Tensor, ...]] = []
                for p in group["params"]:
                    st = self._init_state(p)
                    quads.append((p, p.grad, st["exp_avg"], st["exp_avg_sq"]))
                self._tables[gi] = hip_ops.adam_job_table(quads)
                self._sig[gi] = sigs[gi]
            table = self._tables[gi]
            b1, b2 = group["betas"]
            hip_ops.adam_multi(table, float(group["lr"]), b1, b2, group["eps"], group["weight_decay"], self._host_step[gi])
        self._steps_dirty = True
        return None

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables.clear()
        self._sig.clear()
        self._host_step = [-1] * len(self.param_groups)
        self._steps_dirty = False

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_host_step"):
            self._host_step.append(-1)
