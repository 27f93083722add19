#!/usr/bin/env python
"""Which Python lines issue the small ATen kernels of a train step (fills, copies, adds)?  torch.profiler with stacks."""
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


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    gan, cfg = bench.make_gan(A, dev, "bf16")
    from gan_sr_wind_field_amd.process_data import synthetic_batch
    LR, HR, Z, x, y = (t.to(dev) for t in synthetic_batch(1, 32, 128, 4))
    gan.feed_xy_niter(x, y, torch.tensor(150000, device=dev), 1, 1)
    for i in range(4):
        gan.optimize_parameters(LR, HR, Z, i)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        gan.optimize_parameters(LR, HR, Z, 4)
        gan.optimize_parameters(LR, HR, Z, 5)
        torch.cuda.synchronize()
    want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::zeros",
            "aten::full", "aten::clone", "aten::sum", "aten::to", "aten::_to_copy")
    cnt = Counter()
    for ev in prof.events():
        if ev.name in want:
            frames = [f for f in (ev.stack or []) if "gan_sr_wind_field_amd" in f or "torch/optim" in f or "autograd" in f]
            cnt[(ev.name, frames[0] if frames else "?")] += 1
    for (name, frame), n in cnt.most_common(45):
        print(f"{n:5d} {name:14s} {frame[-110:]}")


if __name__ == "__main__":
    main()
