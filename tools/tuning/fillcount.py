#!/usr/bin/env python
"""Where do the small ATen fill / copy / add kernels of a step come from?  Kernel counts per phase (torch.profiler)."""
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

sys.argv = ["bench.py"]
import bench  # noqa: E402


class A:
    ini, slicing, n, nz, batch, dtype = "local", False, 32, 128, 1, "bf16"


def count(tag, fn):
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    c = Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            n = ev.name
            key = ("fill" if "FillFunctor" in n else "copy" if "copy" in n.lower() else "add" if "Functor_add" in n or "FunctorOnSelf_add" in n
                   else "other_aten" if "at::native" in n else "ours")
            c[key] += 1
    print(f"{tag:34s} {dict(c)}")


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    gan, cfg = bench.make_gan(A, dev, "bf16")
    from gan_sr_wind_field_amd.process_data import synthetic_batch
    LR, HR, Z, x, y = (t.to(dev) for t in synthetic_batch(1, 32, 128, 4))
    gan.feed_xy_niter(x, y, torch.tensor(150000, device=dev), 1, 1)
    for i in range(4):
        gan.optimize_parameters(LR, HR, Z, i)
    gan.G.train()
    out = {}

    def g_fwd():
        out["sr"] = gan.G(LR, Z)

    def g_bwd():
        out["sr"].square().mean().backward()

    count("G forward (train, grad)", g_fwd)
    count("G backward", g_bwd)
    count("optimizer_G.step", gan.optimizer_G.step)
    count("G.zero_grad", lambda: gan.G.zero_grad(set_to_none=True))
    count("G-iteration", lambda: gan.optimize_parameters(LR, HR, Z, 4))
    count("D-iteration", lambda: gan.optimize_parameters(LR, HR, Z, 5))
    count("update_learning_rate", gan.update_learning_rate)


if __name__ == "__main__":
    main()
