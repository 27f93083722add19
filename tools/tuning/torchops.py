#!/usr/bin/env python
"""Which Python lines issue the small ATen kernels of a train step?  A TorchDispatchMode logs every ATen call of one
G-iteration and one D-iteration with the innermost frame of this package (backward runs on the calling thread for it)."""
import os
import sys
import traceback
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

sys.argv = ["bench.py"]
import bench  # noqa: E402


class A:
    ini, slicing, n, nz, batch, dtype = "local", False, 32, 128, 1, "bf16"


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.cnt = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        where = "?"
        for f in reversed(traceback.extract_stack(limit=40)):
            if "gan_sr_wind_field_amd" in f.filename and "tuning" not in f.filename:
                where = f"{os.path.basename(f.filename)}:{f.lineno} {f.name}"
                break
            if "torch/optim" in f.filename:
                where = f"optim/{os.path.basename(f.filename)}:{f.lineno} {f.name}"
        self.cnt[(name, where)] += 1
        return func(*args, **(kwargs or {}))


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.autograd.set_multithreading_enabled(False)
    gan, cfg = bench.make_gan(A, dev, "bf16")
    from gan_sr_wind_field_amd.process_data import synthetic_batch
    LR, HR, Z, x, y = (t.to(dev) for t in synthetic_batch(1, 32, 128, 4))
    gan.feed_xy_niter(x, y, torch.tensor(150000, device=dev), 1, 1)
    for i in range(4):
        gan.optimize_parameters(LR, HR, Z, i)
    torch.cuda.synchronize()
    log = Log()
    with log:
        gan.optimize_parameters(LR, HR, Z, 4)
        gan.optimize_parameters(LR, HR, Z, 5)
        torch.cuda.synchronize()
    skip = ("view", "detach", "alias", "_unsafe_view", "t.default", "expand", "squeeze", "unsqueeze", "slice", "select",
            "permute", "as_strided", "reshape", "transpose", "_local_scalar", "is_", "sym_", "stride", "size")
    tot = 0
    for (name, where), n in log.cnt.most_common():
        if any(s in name for s in skip):
            continue
        tot += n
    print(f"{tot} ATen calls that may launch, two iterations (one G, one D)")
    for (name, where), n in log.cnt.most_common(400):
        if any(s in name for s in skip):
            continue
        print(f"{n:5d} {name:34s} {where}")


if __name__ == "__main__":
    main()
