"""Oracle (CPU, functional) restatement of Generator_3D and Discriminator_3D.

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.

Tensors are logical ``(B, C, X, Y, Z)`` exactly as in the reference; weights are
looked up in a flat mapping under the reference's own ``state_dict`` keys, so a
reference checkpoint (or a golden fixture) can be fed in unchanged.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- #
# hyper-parameter records
# --------------------------------------------------------------------------- #
@dataclass
class GSpec:
    """Generator_3D ctor arguments (reference Generator_3D_Resnet_ESRGAN.py:24-46)."""

    in_channels: int = 4
    out_channels: int = 3
    nf: int = 128
    n_rrdb: int = 16
    upscale: int = 4
    hr_kern: int = 5
    n_rdb_convs: int = 5
    gc: int = 32
    lff_kern: int = 1
    rdb_scale: float = 0.2
    rrdb_scale: float = 0.2
    slope: float = 0.2
    tf: int = 16
    dropout_p: float = 0.0
    n_rdb: int = 3  # RRDB.number_of_RDBs default, torch_blocks.py:308
    #: test aid, not reference behaviour: round every tensor the MI355X bf16 path keeps in HBM (conv operands,
    #: conv / residual outputs and their gradients) to bf16, accumulate in fp32 - the scale of error a bf16
    #: storage format must produce; the bf16 GPU tests derive their per-parameter gradient bounds from it
    bf16_storage: bool = False

    @property
    def n_up(self) -> int:
        # Generator_3D_Resnet_ESRGAN.py:201
        return int(math.floor(math.log2(self.upscale)))


@dataclass
class DSpec:
    """Discriminator_3D ctor arguments (reference Discriminator_3D.py:23-38)."""

    in_channels: int = 3
    bf: int = 32
    feat_kern: int = 3
    nz: int = 10
    enable_slicing: bool = False
    slope: float = 0.2
    dropout_p: float = 0.0
    bn_eps: float = 1e-5
    bn_momentum: float = 0.1
    bf16_storage: bool = False  # test aid, see GSpec.bf16_storage
    norm: str = "batch"         # normalization_type of the blocks: "batch" | "instance" (torch_blocks.py:20-30)


@dataclass
class ConvDesc:
    key: str  # state_dict prefix of the conv ("....0" -> key + ".weight")
    cin: int
    cout: int
    kernel: Tuple[int, int, int]
    stride: Tuple[int, int, int]
    pad: Tuple[int, int, int]
    bn: Optional[str] = None  # state_dict prefix of the BatchNorm3d that follows
    act: bool = True
    inorm: bool = False       # an nn.InstanceNorm3d follows instead (no parameters, no buffers: torch_blocks.py:26-30)


# --------------------------------------------------------------------------- #
# Generator
# --------------------------------------------------------------------------- #
def g_param_shapes(s: GSpec) -> Dict[str, Tuple[int, ...]]:
    """Ordered ``key -> shape`` in the reference's ``state_dict`` order.

    Follows the module construction order of Generator_3D_Resnet_ESRGAN.py:183-222
    and torch_blocks.py:235-283,314-326.
    """
    out: Dict[str, Tuple[int, ...]] = {}
    out["model.0.0.weight"] = (s.nf, s.in_channels, 3, 3, 3)
    for r in range(s.n_rrdb):
        for d in range(s.n_rdb):
            p = f"model.1.module.{r}.RDBs.{d}"
            for i in range(s.n_rdb_convs - 1):
                out[f"{p}.conv{i}.conv.0.weight"] = (s.gc, s.nf + i * s.gc, 3, 3, 3)
            k = s.lff_kern
            out[f"{p}.LFF.weight"] = (s.nf, s.nf + (s.n_rdb_convs - 1) * s.gc, k, k, k)
            out[f"{p}.LFF.bias"] = (s.nf,)
    out[f"model.1.module.{s.n_rrdb}.0.weight"] = (s.nf, s.nf, 3, 3, 3)
    for u in range(s.n_up):
        out[f"model.{2 + u}.1.0.weight"] = (s.nf, s.nf, 3, 3, 3)
    c = s.nf + s.tf
    k = s.hr_kern
    out["hr_convs.0.0.weight"] = (c, c, k, k, k)
    out["hr_convs.2.weight"] = (s.out_channels, c, k, k, k)
    out["hr_convs.2.bias"] = (s.out_channels,)
    out["terrain_convs.0.0.weight"] = (s.tf, 1, 3, 3, 3)
    out["terrain_convs.1.0.weight"] = (s.tf, s.tf, 3, 3, 3)
    return out


def _lrelu(x: Tensor, slope: float) -> Tensor:
    return F.leaky_relu(x, negative_slope=slope)


class _RoundBF16(torch.autograd.Function):
    """bf16 storage emulation (``bf16_storage``): value and gradient both pass through bf16."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def _st(x: Tensor, s) -> Tensor:
    """a tensor the bf16 path stores in HBM"""
    return _RoundBF16.apply(x) if s.bf16_storage else x


def _wq(w: Tensor, s) -> Tensor:
    """compute copy of an fp32 master filter: rounded operand, un-rounded gradient"""
    return w + (w.to(torch.bfloat16).to(w.dtype) - w).detach() if s.bf16_storage else w


def rdb_forward(sd: Dict[str, Tensor], prefix: str, x: Tensor, s: GSpec) -> Tensor:
    """RDB.forward, torch_blocks.py:285-290 (+ RDB_Conv.forward :212-214)."""
    cur = x
    for i in range(s.n_rdb_convs - 1):
        y = F.conv3d(cur, _wq(sd[f"{prefix}.conv{i}.conv.0.weight"], s), None, 1, 1)
        cur = torch.cat((cur, _st(_lrelu(y, s.slope), s)), dim=1)
    pad = (s.lff_kern - 1) // 2
    res = F.conv3d(cur, _wq(sd[f"{prefix}.LFF.weight"], s), sd[f"{prefix}.LFF.bias"], 1, pad)
    return _st(res * s.rdb_scale + x, s)


def rrdb_forward(sd: Dict[str, Tensor], prefix: str, x: Tensor, s: GSpec) -> Tensor:
    """RRDB.forward, torch_blocks.py:328-330."""
    t = x
    for d in range(s.n_rdb):
        t = rdb_forward(sd, f"{prefix}.RDBs.{d}", t, s)
    return _st(t * s.rrdb_scale + x, s)


def generator_trunk(sd: Dict[str, Tensor], x: Tensor, s: GSpec) -> Tensor:
    """``Generator_3D.model`` = feature conv, skip(RRDBs, lr_conv), UpConvs.

    Generator_3D_Resnet_ESRGAN.py:78-94,198,208-220; torch_blocks.py:45-46,345-356.
    """
    f = _st(F.conv3d(_st(x, s), _wq(sd["model.0.0.weight"], s), None, 1, 1), s)
    t = f
    for r in range(s.n_rrdb):
        t = rrdb_forward(sd, f"model.1.module.{r}", t, s)
    t = F.conv3d(t, _wq(sd[f"model.1.module.{s.n_rrdb}.0.weight"], s), None, 1, 1)
    t = _st(f + t, s)
    for u in range(s.n_up):
        t = F.interpolate(t, scale_factor=(2, 2, 1), mode="nearest")
        t = _st(_lrelu(F.conv3d(t, _wq(sd[f"model.{2 + u}.1.0.weight"], s), None, 1, 1), s.slope), s)
    return t


def terrain_features(sd: Dict[str, Tensor], Z: Tensor, s: GSpec) -> Tensor:
    """``Generator_3D.terrain_convs``, Generator_3D_Resnet_ESRGAN.py:120-137."""
    z = _st(_lrelu(F.conv3d(_st(Z, s), _wq(sd["terrain_convs.0.0.weight"], s), None, 1, 1), s.slope), s)
    return _st(F.conv3d(z, _wq(sd["terrain_convs.1.0.weight"], s), None, 1, 1), s)


def generator_forward(
    sd: Dict[str, Tensor],
    x: Tensor,
    Z: Tensor,
    s: GSpec,
    training: bool = False,
    dropout_mask: Optional[Tensor] = None,
) -> Tensor:
    """Generator_3D.forward, Generator_3D_Resnet_ESRGAN.py:225-229.

    ``dropout_mask`` (B, C, 1, 1, 1) of already-scaled keep factors replaces the
    RNG of ``nn.Dropout3d`` (:70-74,104) so CPU and GPU runs can share one mask.
    """
    t = generator_trunk(sd, x, s)
    zf = terrain_features(sd, Z, s)
    h = torch.cat((t, zf), dim=1)
    pad = (s.hr_kern - 1) // 2
    h = _lrelu(F.conv3d(h, _wq(sd["hr_convs.0.0.weight"], s), None, 1, pad), s.slope)
    if dropout_mask is not None:
        h = h * dropout_mask
    elif training and s.dropout_p > 0:
        h = F.dropout3d(h, s.dropout_p, True)
    h = _st(h, s)
    return F.conv3d(h, _wq(sd["hr_convs.2.weight"], s), sd["hr_convs.2.bias"], 1, pad)


# --------------------------------------------------------------------------- #
# Discriminator
# --------------------------------------------------------------------------- #
def d_z_schedule(nz: int) -> List[int]:
    """z-extent after each of the five stages, Discriminator_3D.py:55-64."""
    rem = [nz]
    for i in range(5):
        if i == 0 and nz <= 19:
            rem.append(nz)
        elif i in (1, 2, 3):
            rem.append(rem[i])
        else:
            rem.append(rem[i] // 2 + rem[i] % 2)
    return rem


def d_layers(s: DSpec) -> List[ConvDesc]:
    """Conv/BN/LReLU pyramid of ``Discriminator_3D.features``.

    Discriminator_3D.py:66-169 with create_discriminator_block,
    torch_blocks.py:372-521 (mode "3D" only).
    """
    k = s.feat_kern
    if k not in (3, 5):
        raise NotImplementedError("Only supported kern sizes are 3 and 5")
    p = 2 if k == 5 else 1
    bf = s.bf
    L: List[ConvDesc] = []

    if s.norm not in ("batch", "instance"):
        raise NotImplementedError(f"Unknown norm type {s.norm}")
    inorm = s.norm == "instance"  # (the blocks only: the slicing tail below is always "batch", Discriminator_3D.py:158,167)

    def block(idx: int, cin: int, cout: int, first_norm: bool, halve_z: bool):
        L.append(
            ConvDesc(f"features.{idx}.0.0", cin, cout, (k, k, k), (1, 1, 1), (p, p, p),
                     bn=f"features.{idx}.0.1" if first_norm and not inorm else None, inorm=first_norm and inorm)
        )
        stride = (2, 2, 2) if halve_z else (2, 2, 1)
        L.append(
            ConvDesc(f"features.{idx}.1.0", cout, cout, (4, 4, k), stride, (1, 1, 1),
                     bn=None if inorm else f"features.{idx}.1.1", inorm=inorm)
        )

    block(0, s.in_channels, bf, False, s.nz > 19)
    block(1, bf, 2 * bf, True, False)
    block(2, 2 * bf, 4 * bf, True, False)
    block(3, 4 * bf, 8 * bf, True, False)
    if not s.enable_slicing:
        block(4, 8 * bf, 8 * bf, True, True)
    else:
        # NB the padding of these two layers is the helper's default (1), not p
        # (Discriminator_3D.py:153-169 -> torch_blocks.py:10).
        L.append(ConvDesc("features.4.0", 8 * bf, 8 * bf, (k, k, k), (1, 1, 1), (1, 1, 1),
                          bn="features.4.1"))
        L.append(ConvDesc("features.5.0", 8 * bf, 8 * bf, (k, k, k), (1, 1, 2), (1, 1, 1),
                          bn="features.5.1"))
    return L


def d_param_shapes(s: DSpec) -> Dict[str, Tuple[int, ...]]:
    """Ordered ``key -> shape`` of Discriminator_3D.state_dict()."""
    out: Dict[str, Tuple[int, ...]] = {}
    for l in d_layers(s):
        out[l.key + ".weight"] = (l.cout, l.cin) + tuple(l.kernel)
        if l.bn:
            out[l.bn + ".weight"] = (l.cout,)
            out[l.bn + ".bias"] = (l.cout,)
            out[l.bn + ".running_mean"] = (l.cout,)
            out[l.bn + ".running_var"] = (l.cout,)
            out[l.bn + ".num_batches_tracked"] = ()
    zrem = d_z_schedule(s.nz)[5]
    out["classifier.0.weight"] = (100, 8 * s.bf * 4 * 4 * zrem)
    out["classifier.0.bias"] = (100,)
    out["classifier.2.weight"] = (1, 100)
    out["classifier.2.bias"] = (1,)
    return out


def discriminator_features(
    sd: Dict[str, Tensor], x: Tensor, s: DSpec, training: bool
) -> Tensor:
    """``Discriminator_3D.features``; BN per torch_blocks.py:20-25.

    In training mode the running statistics held in ``sd`` are updated in place,
    exactly like ``nn.BatchNorm3d`` (momentum 0.1, unbiased running variance).
    """
    x = _st(x, s)
    for l in d_layers(s):
        x = F.conv3d(x, _wq(sd[l.key + ".weight"], s), None, l.stride, l.pad)
        if l.bn:
            x = _st(x, s)  # the conv output is stored before the statistics pass
            if training and (l.bn + ".num_batches_tracked") in sd:
                sd[l.bn + ".num_batches_tracked"] += 1
            x = F.batch_norm(
                x,
                sd[l.bn + ".running_mean"],
                sd[l.bn + ".running_var"],
                sd[l.bn + ".weight"],
                sd[l.bn + ".bias"],
                training,
                s.bn_momentum,
                s.bn_eps,
            )
        if l.inorm:  # nn.InstanceNorm3d defaults: no affine, no running statistics, instance statistics in every mode
            x = F.instance_norm(_st(x, s), eps=s.bn_eps)
        if l.act:
            x = _lrelu(x, s.slope)
        x = _st(x, s)
    return x


def discriminator_forward(
    sd: Dict[str, Tensor],
    x: Tensor,
    s: DSpec,
    training: bool = False,
    dropout_mask: Optional[Tensor] = None,
) -> Tensor:
    """Discriminator_3D.forward, Discriminator_3D.py:189-193 -> (B, 1)."""
    h = discriminator_features(sd, x, s, training)
    if dropout_mask is not None:
        h = h * dropout_mask
    elif training and s.dropout_p > 0:
        h = F.dropout3d(h, s.dropout_p, True)
    h = h.reshape(h.shape[0], -1)
    h = F.linear(h, sd["classifier.0.weight"], sd["classifier.0.bias"])
    h = _lrelu(h, s.slope)
    return F.linear(h, sd["classifier.2.weight"], sd["classifier.2.bias"])


# --------------------------------------------------------------------------- #
# weight init (tools/initialization.py:15-34)
# --------------------------------------------------------------------------- #
def kaiming_init_(sd: Dict[str, Tensor], scale: float, gen: torch.Generator) -> None:
    """kaiming_normal(a=0, fan_in) * scale for conv/linear weights, zero biases.

    Same distribution as the reference; *not* the same RNG stream (the reference
    first consumes the default ``nn.Conv3d`` init draws) - golden fixtures carry
    explicit weights wherever stream identity matters.
    """
    bn_prefixes = {k[: -len(".running_mean")] for k in sd if k.endswith(".running_mean")}
    for k, v in sd.items():
        prefix, _, leaf = k.rpartition(".")
        if prefix in bn_prefixes:
            if leaf in ("weight", "running_var"):
                v.fill_(1)
            else:
                v.zero_()
        elif v.dim() >= 2:
            fan_in = v[0].numel()
            std = math.sqrt(2.0 / fan_in)
            v.copy_(torch.randn(v.shape, generator=gen, dtype=v.dtype) * std * scale)
        else:  # conv / linear bias
            v.zero_()


def make_state(shapes: Dict[str, Tuple[int, ...]], dtype=torch.float32) -> Dict[str, Tensor]:
    sd: Dict[str, Tensor] = {}
    for k, shp in shapes.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros(shp, dtype=torch.long)
        else:
            sd[k] = torch.zeros(shp, dtype=dtype)
    return sd


def deterministic_state(
    shapes: Dict[str, Tuple[int, ...]], seed: int, scale: float = 1.0, dtype=torch.float32
) -> Dict[str, Tensor]:
    """Reproducible, *informative* weights keyed like a reference ``state_dict``.

    Used by ``tests/golden/make_golden.py`` (loaded into the real reference
    modules with ``load_state_dict``) and by the tests (fed to the oracle and to
    the HIP modules), so fixtures only need to carry outputs.  Biases, BN affine
    parameters and BN running statistics are perturbed away from their init
    values so that every term of every formula is exercised (a default-init
    Discriminator_3D emits logits ~1e-11).
    """
    sd = make_state(shapes, dtype)
    bn_prefixes = {k[: -len(".running_mean")] for k in sd if k.endswith(".running_mean")}
    for idx, (k, v) in enumerate(sd.items()):
        g = torch.Generator().manual_seed(seed * 100003 + idx)
        prefix, _, leaf = k.rpartition(".")
        if leaf == "num_batches_tracked":
            continue
        if prefix in bn_prefixes:
            if leaf == "weight":
                v.copy_(1.0 + 0.1 * torch.randn(v.shape, generator=g))
            elif leaf == "bias":
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
            elif leaf == "running_mean":
                v.copy_(0.05 * torch.randn(v.shape, generator=g))
            else:
                v.copy_(1.0 + 0.1 * torch.rand(v.shape, generator=g))
        elif v.dim() >= 2:
            std = math.sqrt(2.0 / v[0].numel()) * scale
            v.copy_(torch.randn(v.shape, generator=g) * std)
        else:
            v.copy_(0.01 * torch.randn(v.shape, generator=g))
    return sd
