"""GAN training tricks (reference ``tools/trainingtricks.py:18-59``)."""
import torch


def noisy_labels(label_type: bool, batch_size: int, noise_stddev=torch.tensor(0.05),
                 false_label_val=torch.tensor(0.0), true_label_val=torch.tensor(1.0),
                 val_lower_lim=torch.tensor(0.0), val_upper_lim=torch.tensor(1.0),
                 device: torch.device = torch.device("cpu")) -> torch.Tensor:
    """(batch,) labels = base value + N(0, stddev), clamped to [lower, upper].

    The normal draw happens on the CPU generator even for stddev 0, like the
    reference (:37-39), so the host RNG stream advances identically.
    """
    std = float(noise_stddev)
    noise = torch.normal(mean=0.0, std=torch.full(torch.Size([int(batch_size)]), std))
    base = true_label_val if label_type else false_label_val
    dev = torch.device(device)
    if dev.type == "cuda":
        # no blocking host-to-device copies on the train-step path (a pageable copy waits for the stream to
        # drain): N(0, 0) is exactly zero, a real draw travels through pinned memory, the clamp bounds are scalars
        lo, hi = float(val_lower_lim), float(val_upper_lim)
        if std == 0.0:
            vals = base.to(dev).expand(int(batch_size)) + torch.zeros((), device=dev)
        else:
            vals = noise.pin_memory().to(dev, non_blocking=True) + base
        return vals.clamp(min=lo, max=hi)
    vals = noise.to(device) + base
    return torch.minimum(torch.maximum(vals, val_lower_lim.to(vals.device)), val_upper_lim.to(vals.device))


def instance_noise(sigma_base: torch.Tensor, shape, it: torch.Tensor, niter: torch.Tensor,
                   device=torch.device("cpu")) -> torch.Tensor:
    """Uniform [0,1) noise scaled by sqrt(sigma_base * (1 - (it-1)/niter)) (:49-59).

    (The reference's comment says N(0,1) but it draws ``torch.rand``; NaN once
    it > niter + 1, also as in the reference.)
    """
    scale = torch.sqrt(sigma_base * (1 - (it - 1) / niter))
    if scale.device.type == "cpu" and torch.device(device).type == "cuda":
        # scalars given on the host (wind_field_GAN_3D._noise): the factor is a Python float, draw and scaling are two
        # launches instead of nine (same values: one fp32 multiply per element by the same fp32 factor)
        return torch.rand(shape, device=device).mul_(float(scale))
    return torch.rand(shape, device=device) * scale
