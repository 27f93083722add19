"""Differentiable layer-by-layer evaluation of Generator_3D's sub-stacks on the HIP conv kernels.

The reference's ``G.model`` / ``G.hr_convs`` / ``G.terrain_convs`` are ordinary ``nn.Sequential``s
(reference Generator_3D_Resnet_ESRGAN.py:220-229): a caller may slice them and differentiate through the slice
(plot_data.py:770-793 only runs them forward).  The training path here is the fused program (engine.GeneratorProgram)
- one autograd node for the whole generator.  For a slice called WITH a gradient-requiring input (or with parameters
that require gradients, outside ``torch.no_grad()``) this module evaluates the same containers one layer at a time,
as the reference's modules compose them (torch_blocks.py:192-214 RDB_Conv cat, :278-290 LFF + block residual,
:328-330 RRDB residual, :40-46 skip connection, :345-356 nearest x(2,2,1) + conv + LeakyReLU): every convolution -
forward, input gradient, filter gradient - is a C-ABI launch (``wsr_conv3d_fwd`` / ``_dgrad`` / ``_wgrad``), the glue
(LeakyReLU, concat, residual scaling, nearest up-sampling, Dropout3d) is torch on the same device.  Nothing falls
back to ATen convolutions or to the CPU.  It is an analysis path: correct and differentiable, not fast.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import hip_ops as ops
from .CNN_models import torch_blocks as tb

Tensor = torch.Tensor


def _t3(v):
    return (v, v, v) if isinstance(v, int) else tuple(v)


class _Conv3dFn(torch.autograd.Function):
    """y = conv3d(x, w) + b on the HIP kernels; planar fp32 (B, C, X, Y, Z) in and out, compute dtype ``dt``."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, b: Optional[Tensor], stride, pad, dt):
        x = x.contiguous().float()
        B, cin, X, Y, Z = x.shape
        cout = w.shape[0]
        k = tuple(w.shape[2:])
        cin_p, cout_p = ops.pad_channels(cin, dt), ops.pad_channels(cout, dt)
        wm = w.detach().contiguous().float()
        xb = torch.empty((B, X, Y, Z, cin_p), dtype=dt, device=x.device)
        ops.planar_to_ndhwc(x, xb, 0, cin_p)  # (channels [cin, cin_p) are written as zeros)
        d = ops.make_desc(ops.ConvGeom(cin_p, cout, k, stride, pad), dt, B, (X, Y, Z), cin_p, 0, cout_p, 0)
        yb = torch.zeros((B, d.Xo, d.Yo, d.Zo, cout_p), dtype=dt, device=x.device)
        ops.conv_fwd(d, xb, ops.pack_filter(wm, dt, kpad=cin_p), yb, bias=None if b is None else b.detach().float())
        ctx.save_for_backward(xb, wm)
        ctx.geom = (B, cin, cout, k, tuple(stride), tuple(pad), (X, Y, Z), dt, b is not None, cin_p, cout_p)
        return ops.ndhwc_to_planar(yb, cout)

    @staticmethod
    def backward(ctx, g: Tensor):
        xb, wm = ctx.saved_tensors
        B, cin, cout, k, stride, pad, xyz, dt, has_b, cin_p, cout_p = ctx.geom
        g = g.contiguous().float()
        gb = torch.empty(tuple(g.shape[:1]) + tuple(g.shape[2:]) + (cout_p,), dtype=dt, device=g.device)
        ops.planar_to_ndhwc(g, gb, 0, cout_p)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dd = ops.make_desc(ops.ConvGeom(cin, cout_p, k, stride, pad), dt, B, xyz, cin, 0, cout_p, 0)
            dx = torch.zeros((B, cin) + tuple(xyz), dtype=torch.float32, device=g.device)
            ops.conv_dgrad(dd, gb, ops.pack_filter(wm, dt, transpose=True, kpad=cout_p), dx, dx_planar=True)
        if ctx.needs_input_grad[1]:
            d = ops.make_desc(ops.ConvGeom(cin_p, cout, k, stride, pad), dt, B, xyz, cin_p, 0, cout_p, 0)
            dwp = torch.zeros((cout, k[0] * k[1] * k[2], cin_p), dtype=torch.float32, device=g.device)
            ops.conv_wgrad(d, xb, gb, dwp)
            dw = torch.zeros_like(wm)
            ops.unpack_wgrad(dwp, dw, 1.0, accumulate=False)
        if has_b and ctx.needs_input_grad[2]:
            db = g.sum(dim=(0, 2, 3, 4))
        return dx, dw, db, None, None, None


def conv3d(x: Tensor, conv: nn.Conv3d, dt: torch.dtype) -> Tensor:
    if conv.dilation != (1, 1, 1) or conv.groups != 1 or conv.padding_mode != "zeros":
        raise NotImplementedError("the HIP conv kernels take dense, undilated, zero-padded 3D convolutions")
    return _Conv3dFn.apply(x, conv.weight, conv.bias, _t3(conv.stride), _t3(conv.padding), dt)


def run(m: nn.Module, x: Tensor, dt: torch.dtype) -> Tensor:
    """``m(x)`` as the reference's module of the same class computes it, convolutions on the HIP kernels."""
    if isinstance(m, nn.Conv3d):
        return conv3d(x, m, dt)
    if isinstance(m, nn.LeakyReLU):
        return F.leaky_relu(x, m.negative_slope)
    if isinstance(m, nn.Upsample):
        return F.interpolate(x, scale_factor=m.scale_factor, mode=m.mode)
    if isinstance(m, (nn.Dropout3d, nn.Dropout, nn.Identity)):
        return m(x)
    if isinstance(m, tb.SkipConnectionBlock):        # reference torch_blocks.py:40-46
        return x + run(m.module, x, dt)
    if isinstance(m, tb.RRDB):                       # :328-330
        return run(m.RDBs, x, dt) * m.RRDB_residual_scaling + x
    if isinstance(m, tb.RDB):                        # :278-290
        t = x
        for i in range(m.number_of_convs):
            t = run(getattr(m, f"conv{i}"), t, dt)
        return run(m.LFF, t, dt) * m.residual_scaling + x
    if isinstance(m, tb.RDB_Conv):                   # :202-214
        return torch.cat((x, run(m.conv, x, dt)), dim=1)
    if isinstance(m, nn.Sequential):
        for child in m:
            x = run(child, x, dt)
        return x
    raise NotImplementedError(f"no layer-wise HIP evaluation for {type(m).__name__}")
