"""Generator_3D: ESRGAN-style RRDB generator for 3-D wind fields on MI355X.

Same constructor signature, attribute names (``model``, ``hr_convs``,
``terrain_convs``, ``max_norm``) and ``state_dict`` keys as the reference
(CNN_models/Generator_3D_Resnet_ESRGAN.py:23-229); the sub-modules are built in
the reference's order so a seeded construction + ``init_weights`` reproduces the
reference's initial weights.  ``forward(x, Z)`` runs the fused HIP program
(engine.GeneratorProgram); there is no ATen convolution and no CPU path.

``use_mixed_precision`` - parsed but unused by the reference (its AMP lines are
commented out, :65) - selects the compute dtype here: False -> fp32 MFMA (exact
fp32 products/accumulation), True -> bf16 operands with fp32 accumulation and an
fp32 output / loss.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import engine
from ..tools import loggingclass as lc
from .torch_blocks import RRDB, SkipConnectionBlock, create_conv_lrelu_layer, create_UpConv_block


class ProgramStack(nn.Sequential):
    """``Generator_3D.model`` / ``.hr_convs`` / ``.terrain_convs``: the reference's ``nn.Sequential`` (same children,
    same ``state_dict`` keys) whose elements are executed by the generator's HIP program.  Calling the stack, or a
    slice of it - ``G.model[:2](LR)``, ``G.hr_convs[:-2](t)``, ``G.terrain_convs(Z)`` as in the reference's
    ``plot_data.py:770-793`` - runs those elements one after the other on planar fp32 tensors: under ``torch.no_grad()``
    through the fused program's stages, with gradients enabled layer by layer on the HIP conv kernels (``layerwise.py``:
    differentiable like the reference's ``nn.Sequential``); the training path is ``Generator_3D.forward``."""

    def bind(self, owner: "Generator_3D", name: str, lo: int = 0):
        import weakref
        object.__setattr__(self, "_owner", weakref.ref(owner))  # (not a sub-module: no cycle in .modules())
        object.__setattr__(self, "_stack", name)
        object.__setattr__(self, "_lo", lo)
        return self

    def __getstate__(self):  # torch.save(G) / copy.deepcopy(G): a weak reference does not pickle - the owner rebinds
        d = dict(self.__dict__)
        d.pop("_owner", None)
        return d

    def __getitem__(self, idx):
        if not isinstance(idx, slice):
            return super().__getitem__(idx)
        n = len(self)
        lo, hi, step = idx.indices(n)
        if step != 1:
            raise IndexError("slices of a program stack are contiguous")
        sub = ProgramStack(*[super(ProgramStack, self).__getitem__(i) for i in range(lo, hi)])
        return sub.bind(self._owner(), self._stack, self._lo + lo)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        owner = getattr(self, "_owner", lambda: None)()
        if owner is None:
            raise RuntimeError("this stack is not bound to a Generator_3D")
        if not x.is_cuda:
            raise RuntimeError("Generator_3D runs on the MI355X HIP kernels only (no CPU fallback)")
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            # the reference's nn.Sequential is differentiable here (Generator_3D_Resnet_ESRGAN.py:220-229): so is this
            # one - layer by layer, every convolution (forward, input gradient, filter gradient) a HIP launch
            # (layerwise.py); under torch.no_grad() / on frozen parameters the fused program below runs instead
            from .. import layerwise
            for child in self:
                x = layerwise.run(child, x, owner.compute_dtype)
            return x
        prog = owner.program()
        with torch.no_grad():
            for k, child in enumerate(self):
                if isinstance(child, (nn.Dropout3d, nn.Dropout)):
                    x = child(x)  # (hr_convs[1]: identity in eval mode, torch's own mask otherwise)
                else:
                    x = prog.run_stage(self._stack, self._lo + k, x)
        return x


class Generator_3D(nn.Module, lc.GlobalLoggingClass):
    def __init__(self, in_channels: int, out_channels: int, number_of_features: int, number_of_RRDBs: int,
                 upscale: int = 4, hr_kern_size: int = 3, number_of_RDB_convs: int = 5, RDB_gc: int = 32,
                 lff_kern_size: int = 1, RDB_residual_scaling: float = 0.2, RRDB_residual_scaling: float = 0.2,
                 act_type: str = "leakyrelu", number_of_z_layers: int = 10, conv_mode: str = "3D",
                 use_mixed_precision: bool = False, device="cpu", terrain_number_of_features: int = 16,
                 dropout_probability: float = 0.0, max_norm: float = 1.0):
        super().__init__()
        if act_type == "leakyrelu":
            slope = 0.2
        elif act_type == "relu":
            slope = 0.0
        else:
            self.status_logs.append(f"Generator: warning: activation type {act_type} has not been implemented "
                                    "- defaulting to leaky ReLU (0.2)")
            slope = 0.2
        if conv_mode != "3D":
            if conv_mode in ("2D", "horizontal_3D"):
                raise NotImplementedError(f"conv_mode {conv_mode}: only 3D runs on the MI355X path")
            raise ValueError(f"Conv mode {conv_mode} not implemented")
        self.slope = slope
        self.max_norm = max_norm
        self.compute_dtype = engine.compute_dtype_of(use_mixed_precision)
        nf, tf = number_of_features, terrain_number_of_features
        hr_pad = (hr_kern_size - 1) // 2
        dropout = nn.Dropout3d(p=0.0 if dropout_probability is None else dropout_probability)

        # construction order == reference order (RNG parity of the default nn.Conv3d init)
        feature_conv = create_conv_lrelu_layer(in_channels, nf, 3, padding=1, lrelu=False)
        lr_conv = create_conv_lrelu_layer(nf, nf, 3, padding=1, lrelu_negative_slope=slope, lrelu=False)
        hr_convs = [
            create_conv_lrelu_layer(nf + tf, nf + tf, kernel_size=hr_kern_size, padding=hr_pad,
                                    lrelu_negative_slope=slope),
            dropout,
            nn.Conv3d(nf + tf, out_channels, kernel_size=hr_kern_size, padding=hr_pad),
        ]
        terrain_convs = [
            create_conv_lrelu_layer(1, tf, 3, padding=1, lrelu=True),
            create_conv_lrelu_layer(tf, tf, 3, padding=1, lrelu=False),
        ]
        rrdbs = [RRDB(nf, RDB_gc, number_of_RDB_convs, lff_kern_size, lrelu_negative_slope=slope,
                      RDB_residual_scaling=RDB_residual_scaling, RRDB_residual_scaling=RRDB_residual_scaling,
                      mode=conv_mode) for _ in range(number_of_RRDBs)]
        shortcut = SkipConnectionBlock(nn.Sequential(*rrdbs, lr_conv))
        n_up = math.floor(math.log2(upscale))
        if 2 ** n_up != upscale:
            self.status_logs.append(f"ESRDnet: warning: upsampling only supported for factors 2^n. "
                                    f"Defaulting {upscale} to {2 ** n_up}")
        upsampler = [create_UpConv_block(nf, nf, scale=2, lrelu_negative_slope=slope,
                                         number_of_z_layers=number_of_z_layers, mode=conv_mode)
                     for _ in range(n_up)]
        self.model = ProgramStack(feature_conv, shortcut, *upsampler).bind(self, "model")
        self.hr_convs = ProgramStack(*hr_convs).bind(self, "hr_convs")
        self.terrain_convs = ProgramStack(*terrain_convs).bind(self, "terrain_convs")
        self._program = None
        self.status_logs.append("Generator: finished init")

    def train(self, mode: bool = True):
        """``nn.Module.train`` walks all ~2 000 sub-modules (about 1.5 ms of host time per call, and the
        train step toggles G three times); the sub-modules here are parameter containers whose forward is
        never called, so after a first full pass only the flags that are read are kept current."""
        if getattr(self, "_flags_synced", None) is None:
            super().train(mode)
            self._flags_synced = mode
            return self
        self.training = mode
        self.hr_convs.training = mode
        self.hr_convs[1].training = mode  # Dropout3d
        if self._flags_synced != mode:
            self._flags_dirty = True
        return self

    def sync_module_flags(self):
        """bring every sub-module's ``training`` flag up to date (``modules()`` consumers, state dumps)"""
        nn.Module.train(self, self.training)
        self._flags_synced = self.training
        return self

    def program(self) -> "engine.GeneratorProgram":
        if self._program is None or self._program.dt != self.compute_dtype:
            self._program = engine.GeneratorProgram(self, self.compute_dtype)
        return self._program

    def forward(self, x: torch.Tensor, Z: torch.Tensor, dropout_scale: torch.Tensor = None) -> torch.Tensor:
        """(B, Cin, n, n, nz), (B, 1, s*n, s*n, nz) -> (B, Cout, s*n, s*n, nz) fp32."""
        if not x.is_cuda:
            raise RuntimeError("Generator_3D runs on the MI355X HIP kernels only: move the model and its inputs to "
                               "a cuda device (there is no CPU fallback)")
        return engine.run_generator(self.program(), x, Z, self.training, dropout_scale)

    def __getstate__(self):  # torch.save(G): the program holds device workspaces and ctypes handles - it is rebuilt
        d = dict(self.__dict__)
        d["_program"] = None
        return d

    def __setstate__(self, state):
        self.__dict__.update(state)
        for name in ("model", "hr_convs", "terrain_convs"):
            getattr(self, name).bind(self, name)

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, None if k == "_program" else copy.deepcopy(v, memo))
        for name in ("model", "hr_convs", "terrain_convs"):  # (the copies' stacks belong to the copy)
            getattr(new, name).bind(new, name)
        return new
