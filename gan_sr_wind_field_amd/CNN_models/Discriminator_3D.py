"""Discriminator_3D: VGG-style 3-D discriminator on MI355X.

Constructor signature, attributes (``features``, ``classifier``, ``dropout``) and
``state_dict`` keys follow the reference (CNN_models/Discriminator_3D.py:15-193).
``features`` is an ``nn.Sequential`` subclass whose forward runs the fused HIP
program (conv + BatchNorm3d + LeakyReLU pyramid); it stays deep-copyable and
callable on its own because the reference uses ``copy.deepcopy(D.features)`` as
a perceptual feature extractor (wind_field_GAN_3D.py:577-583).  The classifier
head (two tiny Linear layers on 256*4*4*z features) stays on rocBLAS via
``torch.nn.Linear``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import engine
from ..tools import loggingclass as lc
from .torch_blocks import create_conv_lrelu_layer, create_discriminator_block


class FeaturePyramid(nn.Sequential):
    """``Discriminator_3D.features``: same children / keys as the reference's
    ``nn.Sequential``, executed as one fused program."""

    slope: float = 0.2
    compute_dtype: torch.dtype = torch.float32
    _program = None

    def _layers(self):
        out = []

        def add(prefix, seq):
            conv = seq[0]
            bn = seq[1] if len(seq) > 1 and isinstance(seq[1], (nn.BatchNorm3d, nn.InstanceNorm3d)) else None
            act = isinstance(seq[-1], nn.LeakyReLU)
            out.append(engine.DLayer(engine.site_from_conv(prefix + ".0", conv), bn, act))

        for i, child in enumerate(self):
            if isinstance(child[0], nn.Sequential):  # discriminator block: two conv groups
                for j, grp in enumerate(child):
                    add(f"features.{i}.{j}", grp)
            else:  # plain conv-BN-LReLU group (enable_slicing tail)
                add(f"features.{i}", child)
        return out

    def program(self) -> "engine.DiscriminatorProgram":
        if self._program is None or self._program.dt != self.compute_dtype:
            self._program = engine.DiscriminatorProgram(self._layers(), self.slope, self.compute_dtype)
        return self._program

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("Discriminator_3D runs on the MI355X HIP kernels only (no CPU fallback)")
        return engine.run_discriminator_features(self.program(), x, self.training)

    def forward_pair(self, xa: torch.Tensor, xb: torch.Tensor) -> torch.Tensor:
        """features of two equally shaped inputs in one batched pass, stacked along the batch axis: the
        convolutions run once on both, BatchNorm keeps per-call statistics (training mode: xa's call first)"""
        if not (xa.is_cuda and xb.is_cuda):
            raise RuntimeError("Discriminator_3D runs on the MI355X HIP kernels only (no CPU fallback)")
        return engine.run_discriminator_features_pair(self.program(), xa, xb, self.training)

    def __deepcopy__(self, memo):
        import copy
        new = FeaturePyramid(*[copy.deepcopy(m, memo) for m in self])
        new.slope, new.compute_dtype = self.slope, self.compute_dtype
        new.train(self.training)
        return new


class Discriminator_3D(nn.Module, lc.GlobalLoggingClass):
    def __init__(self, in_channels: int, base_number_of_features: int, feat_kern_size: int = 3,
                 normalization_type: str = "batch", act_type: str = "leakyrelu", mode="CNA", device="cpu",
                 number_of_z_layers=10, conv_mode: str = "3D", use_mixed_precision: bool = False,
                 enable_slicing: bool = False, dropout_probability: float = 0.0):
        super().__init__()
        self.base_number_of_features = bf = base_number_of_features
        if act_type == "leakyrelu":
            slope = 0.2
        elif act_type == "relu":
            slope = 0.0
        else:
            self.status_logs.append(f"Discriminator: warning: activation type {act_type} has not been implemented "
                                    "- defaulting to leaky ReLU (0.2)")
            slope = 0.2
        if conv_mode != "3D":
            raise NotImplementedError(f"conv_mode {conv_mode}: only 3D runs on the MI355X path")

        # z extent after each stage (reference :55-64): stage 0 halves z only when nz > 19,
        # stages 1-3 keep it, stage 4 halves (rounding up)
        zrem = [number_of_z_layers]
        for i in range(5):
            if i == 0 and number_of_z_layers <= 19:
                zrem.append(number_of_z_layers)
            elif i in (1, 2, 3):
                zrem.append(zrem[i])
            else:
                zrem.append(zrem[i] // 2 + zrem[i] % 2)

        def block(cin, cout, first_norm, halve_z, nz):
            return create_discriminator_block(cin, cout, feat_kern_size=feat_kern_size, lrelu_negative_slope=slope,
                                              normalization_type=normalization_type,
                                              drop_first_norm=not first_norm, halve_z_dim=halve_z,
                                              number_of_z_layers=nz, mode=conv_mode)

        feats = [
            block(in_channels, bf, False, number_of_z_layers > 19, zrem[0]),
            block(bf, bf * 2, True, False, zrem[1]),
            block(bf * 2, bf * 4, True, False, zrem[2]),
            block(bf * 4, bf * 8, True, False, zrem[3]),
        ]
        if not enable_slicing:
            feats.append(block(bf * 8, bf * 8, True, True, zrem[4]))
        else:
            feats.append(create_conv_lrelu_layer(bf * 8, bf * 8, feat_kern_size, normalization_type="batch"))
            feats.append(create_conv_lrelu_layer(bf * 8, bf * 8, feat_kern_size, stride=(1, 1, 2),
                                                 normalization_type="batch"))
        classifier = [nn.Linear(bf * 8 * 4 * 4 * zrem[5], 100), nn.LeakyReLU(negative_slope=slope),
                      nn.Linear(100, 1)]
        self.dropout = nn.Dropout3d(p=dropout_probability)
        self.features = FeaturePyramid(*feats)
        self.features.slope = slope
        self.features.compute_dtype = engine.compute_dtype_of(use_mixed_precision)
        self.classifier = nn.Sequential(*classifier)
        self.status_logs.append("Discriminator: finished init")

    @property
    def compute_dtype(self) -> torch.dtype:
        return self.features.compute_dtype

    @compute_dtype.setter
    def compute_dtype(self, dt) -> None:
        self.features.compute_dtype = engine.compute_dtype_of(dt)

    def _classify(self, h: torch.Tensor) -> torch.Tensor:
        """``self.classifier(h)``; its first layer - a few samples x 100 outputs x ~10^5 features, a weight matrix of
        tens of MB read once - goes through the streaming row kernel (``wsr_linear_rows``) where that applies"""
        from .. import hip_ops
        lin = self.classifier[0]
        z = hip_ops.linear_rows(h, lin.weight, lin.bias) if isinstance(lin, nn.Linear) else None
        if z is None:
            return self.classifier(h)
        for m in list(self.classifier)[1:]:
            z = m(z)
        return z

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = self.dropout(self.features(x))
        h = h.reshape(h.shape[0], -1)  # logical (C, X, Y, Z) order, as in the reference (:191-192)
        return self._classify(h)

    def forward_pair(self, xa: torch.Tensor, xb):
        """(D(xa), D(xb)) as two consecutive calls would give them - the reference's D(real), D(fake) of one iteration
        (wind_field_GAN_3D.py:247-304) - with the feature pyramid run once on both inputs (see
        ``FeaturePyramid.forward_pair``).  ``xb`` may be a callable that builds the second input (the reference draws
        the fake sample's instance noise AFTER the first call): the random draws keep the reference's order - first
        call's Dropout3d mask, ``xb()``, second call's mask; a Dropout3d mask is one Bernoulli draw per (sample,
        channel), whatever the spatial extent, so it is drawn on a (B, C, 1, 1, 1) tensor ahead of the features."""
        if not xa.is_cuda:
            raise RuntimeError("Discriminator_3D runs on the HIP kernels only (no CPU fallback): move the inputs "
                               "to the MI355X device")
        b = xa.shape[0]
        c = self.features.program().layers[-1].conv.cout
        drop = self.training and self.dropout.p > 0
        ones = (lambda: torch.ones((b, c, 1, 1, 1), dtype=torch.float32, device=xa.device)) if drop else None
        mask_a = self.dropout(ones()) if drop else None
        if callable(xb):
            xb = xb()
        mask_b = self.dropout(ones()) if drop else None
        f = self.features.forward_pair(xa, xb)
        outs = []
        for h, m in ((f[:b], mask_a), (f[b:], mask_b)):
            if m is not None:
                h = h * m
            outs.append(self._classify(h.reshape(h.shape[0], -1)))
        return outs[0], outs[1]
