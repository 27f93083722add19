"""Wind-field derivative operators used by the generator loss.

Only the hot-path part of the reference's ``process_data.py`` lives here
(:273-313); dataset I/O and augmentation are outside this package's scope.
Tensors are logical (B, C, X, Y, Z); everything is differentiable torch code
that runs on whatever device its inputs are on (fp32 loss math).
"""
import torch


def calculate_div_z(HR_data: torch.Tensor, Z: torch.Tensor) -> torch.Tensor:
    """d/dz on the non-uniform terrain-following levels ``Z`` (B, 1, X, Y, nz).

    Second-order three-point stencil inside, one-sided first differences at the
    lowest / highest level (reference :273-298).
    """
    f = HR_data
    h = Z[..., 1:] - Z[..., :-1]
    below, above = h[..., :-1], h[..., 1:]
    inner = (below ** 2 * f[..., 2:] + (above ** 2 - below ** 2) * f[..., 1:-1] - above ** 2 * f[..., :-2]) / (
        below * above * (below + above))
    bottom = (f[..., 1:2] - f[..., 0:1]) / h[..., 0:1]
    top = (f[..., -1:] - f[..., -2:-1]) / h[..., -1:]
    return torch.cat((bottom, inner, top), dim=-1)


def calculate_gradient_of_wind_field(HR_data: torch.Tensor, x: torch.Tensor, y: torch.Tensor,
                                     Z: torch.Tensor) -> torch.Tensor:
    """(B, 3, X, Y, nz) -> (B, 9, X, Y, nz): d/dx, d/dy (coordinate spacing), d/dz (reference :301-313)."""
    grad_x, grad_y = torch.gradient(HR_data, dim=(2, 3), spacing=(x, y))
    return torch.cat((grad_x, grad_y, calculate_div_z(HR_data, Z)), dim=1)
