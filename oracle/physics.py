"""Oracle restatement of the wind-field derivative operators and metrics.

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import torch

Tensor = torch.Tensor


def ddz_terrain_following(f: Tensor, Z: Tensor) -> Tensor:
    """Vertical derivative on the non-uniform terrain-following levels.

    Follows process_data.py:273-298 (``calculate_div_z``): three-point formula in
    the interior, one-sided first differences at the bottom and top level.
    ``f`` is (B, C, X, Y, nz); ``Z`` is (B, 1, X, Y, nz).
    """
    h = Z[..., 1:] - Z[..., :-1]  # (B,1,X,Y,nz-1), broadcast over channels
    lo, hi = h[..., :-1], h[..., 1:]  # spacing below / above an interior level
    out = torch.zeros_like(f)
    out[..., 1:-1] = (
        lo**2 * f[..., 2:] + (hi**2 - lo**2) * f[..., 1:-1] - hi**2 * f[..., :-2]
    ) / (lo * hi * (lo + hi))
    out[..., -1] = (f[..., -1] - f[..., -2]) / h[..., -1]
    out[..., 0] = (f[..., 1] - f[..., 0]) / h[..., 0]
    return out


def ddcoord(f: Tensor, c: Tensor, dim: int) -> Tensor:
    """``torch.gradient(f, dim=dim, spacing=(c,))`` written out (edge_order=1).

    Interior points use the second-order non-uniform central formula, the two
    edges first-order one-sided differences - what process_data.py:303 relies on.
    """
    f = f.movedim(dim, -1)
    hl = (c[1:-1] - c[:-2]).to(f.dtype)
    hr = (c[2:] - c[1:-1]).to(f.dtype)
    out = torch.empty_like(f)
    out[..., 1:-1] = (
        hl**2 * f[..., 2:] - hr**2 * f[..., :-2] + (hr**2 - hl**2) * f[..., 1:-1]
    ) / (hl * hr * (hl + hr))
    out[..., 0] = (f[..., 1] - f[..., 0]) / (c[1] - c[0]).to(f.dtype)
    out[..., -1] = (f[..., -1] - f[..., -2]) / (c[-1] - c[-2]).to(f.dtype)
    return out.movedim(-1, dim)


def wind_gradient(uvw: Tensor, x: Tensor, y: Tensor, Z: Tensor) -> Tensor:
    """9-channel Jacobian stack, process_data.py:301-313.

    Channel order: d(u,v,w)/dx, d(u,v,w)/dy, d(u,v,w)/dz.
    """
    return torch.cat(
        (ddcoord(uvw, x, 2), ddcoord(uvw, y, 3), ddz_terrain_following(uvw, Z)), dim=1
    )


def gradient_norm_factors(g_hr: Tensor, g_sr: Tensor) -> List[Tensor]:
    """Batch-global normalisers, wind_field_GAN_3D.py:773-814.

    NB the z-gradient maximum is taken WITHOUT ``abs`` (:780-781), as in the
    reference.  Returns [xy_gradient, z_gradient, divergence, xy_divergence].
    """

    def div3(g):
        return g[:, 0] + g[:, 4] + g[:, 8]

    def div2(g):
        return g[:, 0] + g[:, 4]

    pairs = [
        (g_hr[:, :6].abs().max(), g_sr[:, :6].abs().max()),
        (g_hr[:, 6:].max(), g_sr[:, 6:].max()),
        (div3(g_hr).abs().max(), div3(g_sr).abs().max()),
        (div2(g_hr).abs().max(), div2(g_sr).abs().max()),
    ]
    return [torch.max(a, b / 100) for a, b in pairs]


def psnr(HR: Tensor, SR: Tensor, max_diff_squared: float = 4.0, eps: float = 1e-8) -> Tensor:
    """wind_field_GAN_3D.py:730-742 (MSE averaged over B*X*Y*Z, *not* channels)."""
    w, h, l = HR.shape[2], HR.shape[3], HR.shape[4]
    mse = torch.sum((HR - SR) ** 2) / (w * h * l * HR.shape[0])
    return torch.tensor(10.0) * math.log10(max_diff_squared / (float(mse) + eps))


def trilinear_baseline(LR: Tensor, scale: int) -> Tensor:
    """wind_field_GAN_3D.py:759-764: trilinear, align_corners, first 3 channels."""
    return torch.nn.functional.interpolate(
        LR[:, :3], scale_factor=(scale, scale, 1), mode="trilinear", align_corners=True
    )
