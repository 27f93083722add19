"""Building blocks of Generator_3D / Discriminator_3D (3D conv mode).

Public names, constructor arguments, sub-module names and therefore
``state_dict`` keys follow the reference's ``CNN_models/torch_blocks.py``
(:5-47, :192-369, :372-521).  The blocks are *parameter containers with a
recipe*: the arithmetic of the hot path is executed by the fused HIP programs
in ``engine.py`` (dense-block buffers instead of ``cat``, LeakyReLU / residual /
up-sampling folded into the conv kernels), which walk these containers.  The
2D and ``horizontal_3D`` modes of the reference are experimental there and
unused by every shipped config; they raise ``NotImplementedError`` here.
"""
from __future__ import annotations

from typing import Tuple, Union

import torch
from torch import nn

Int3 = Union[int, Tuple[int, int, int]]


def _triple(v: Int3) -> Tuple[int, int, int]:
    return (v, v, v) if isinstance(v, int) else tuple(v)


def _only_3d(layer_type) -> None:
    if layer_type is not nn.Conv3d:
        raise NotImplementedError("only the 3D conv mode is implemented on the MI355X path")


def create_conv_lrelu_layer(in_channels, out_channels, kernel_size, stride=1, padding=1,
                            lrelu_negative_slope=0.2, normalization_type="", layer_type=nn.Conv3d,
                            lrelu=True) -> nn.Sequential:
    """``Sequential(Conv3d(bias=False) [, BatchNorm3d] [, LeakyReLU])`` - index 0 is
    always the conv, so keys end in ``.0.weight`` (reference :5-37)."""
    _only_3d(layer_type)
    layers = [nn.Conv3d(in_channels, out_channels, kernel_size, stride, padding, bias=False)]
    if normalization_type:
        if normalization_type == "batch":
            layers.append(nn.BatchNorm3d(out_channels))
        elif normalization_type == "instance":  # (reference :26-30: no affine parameters, no running statistics)
            layers.append(nn.InstanceNorm3d(out_channels))
        else:
            raise NotImplementedError(f"Unknown norm type {normalization_type}")
    if lrelu:
        layers.append(nn.LeakyReLU(negative_slope=lrelu_negative_slope))
    return nn.Sequential(*layers)


class _Recipe(nn.Module):
    """A block that is executed by a fused program, never layer by layer."""

    def forward(self, *args, **kwargs):  # pragma: no cover - guard
        raise RuntimeError(
            f"{type(self).__name__} is executed by the fused HIP program of its parent network "
            "(Generator_3D / Discriminator_3D.features); call the network, not the block")


class SkipConnectionBlock(_Recipe):
    """``x + module(x)`` (reference :40-46); the add is the lr_conv epilogue."""

    def __init__(self, submodule: nn.Module):
        super().__init__()
        self.module = submodule


class RDB_Conv(_Recipe):
    """conv k3 + LeakyReLU whose output is appended to its input (reference :192-214):
    the kernel writes into the next channel window of the dense-block buffer."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3,
                 lrelu_negative_slope: float = 0.2, layer_type=nn.Conv3d):
        super().__init__()
        self.conv = create_conv_lrelu_layer(in_channels, out_channels, kernel_size, stride=1,
                                            padding=(kernel_size - 1) // 2,
                                            lrelu_negative_slope=lrelu_negative_slope, layer_type=layer_type)


class RDB(_Recipe):
    """Residual dense block (reference :217-290): ``x + scale * LFF(dense(x))``."""

    def __init__(self, in_channels: int, growth_channels: int, number_of_conv_layers: int,
                 lff_kern_size: int = 1, lrelu_negative_slope: float = 0.2, residual_scaling=0.2, mode="2D"):
        super().__init__()
        if mode != "3D":
            raise NotImplementedError(f"RDB mode {mode}: only 3D is implemented on the MI355X path")
        self.residual_scaling = residual_scaling
        self.number_of_convs = number_of_conv_layers - 1
        for i in range(self.number_of_convs):
            self.add_module(f"conv{i}", RDB_Conv(in_channels + i * growth_channels, growth_channels,
                                                 lrelu_negative_slope=lrelu_negative_slope,
                                                 layer_type=nn.Conv3d))
        if lff_kern_size <= 0 or lff_kern_size % 2 == 0:
            raise ValueError("LFF kernel size (lff_kern_size) must be an odd number > 0")
        self.LFF = nn.Conv3d(in_channels + self.number_of_convs * growth_channels, in_channels,
                             kernel_size=lff_kern_size, padding=(lff_kern_size - 1) // 2)


class RRDB(_Recipe):
    """Residual-in-residual dense block (reference :293-330)."""

    def __init__(self, in_channels: int, growth_channels: int, num_convs: int, lff_kern_size: int = 1,
                 lrelu_negative_slope: float = 0.2, RDB_residual_scaling: float = 0.2,
                 RRDB_residual_scaling: float = 0.2, number_of_RDBs: int = 3, mode="2D"):
        super().__init__()
        self.RRDB_residual_scaling = RRDB_residual_scaling
        self.RDBs = nn.Sequential(*[
            RDB(in_channels, growth_channels, num_convs, lrelu_negative_slope=lrelu_negative_slope,
                residual_scaling=RDB_residual_scaling, lff_kern_size=lff_kern_size, mode=mode)
            for _ in range(number_of_RDBs)])


def create_UpConv_block(in_channels: int, out_channels: int, scale: int, lrelu_negative_slope: float = 0.2,
                        mode="2D", number_of_z_layers=10) -> nn.Sequential:
    """nearest x(scale, scale, 1) -> conv k3 -> LeakyReLU (reference :333-356); the
    up-sampling is folded into the conv's input gather."""
    if mode != "3D":
        raise NotImplementedError(f"Unknown / unsupported UpConv mode {mode}")
    return nn.Sequential(
        nn.Upsample(scale_factor=(scale, scale, 1), mode="nearest"),
        create_conv_lrelu_layer(in_channels, out_channels, kernel_size=3, padding=1,
                                lrelu_negative_slope=lrelu_negative_slope, layer_type=nn.Conv3d))


def create_discriminator_block(in_channels: int, out_channels: int, feat_kern_size: int = 3,
                               lrelu_negative_slope: float = 0.2, normalization_type: str = "batch",
                               drop_first_norm: bool = False, mode: str = "2D", number_of_z_layers: int = 10,
                               halve_z_dim: bool = True) -> nn.Sequential:
    """[conv k s1 (+BN) + LReLU] -> [conv (4,4,k) s(2,2,1|2) p1 + BN + LReLU] (reference :372-521)."""
    if feat_kern_size not in (3, 5):
        raise NotImplementedError("Only supported kern sizes are 3 and 5")
    if mode != "3D":
        raise NotImplementedError("Only mode 3D is implemented on the MI355X path")
    pad = 2 if feat_kern_size == 5 else 1
    first = create_conv_lrelu_layer(in_channels, out_channels, kernel_size=feat_kern_size,
                                    lrelu_negative_slope=lrelu_negative_slope, padding=pad, stride=1,
                                    normalization_type="" if drop_first_norm else normalization_type)
    down = create_conv_lrelu_layer(out_channels, out_channels, kernel_size=(4, 4, feat_kern_size),
                                   lrelu_negative_slope=lrelu_negative_slope,
                                   padding=1 if halve_z_dim else (1, 1, 1),
                                   stride=2 if halve_z_dim else (2, 2, 1),
                                   normalization_type=normalization_type)
    return nn.Sequential(first, down)
