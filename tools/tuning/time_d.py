#!/usr/bin/env python
"""Discriminator alone at the benchmark's size (128^3, bf 32, bf16): device time of the four kinds of pass a train
step issues.  python tools/tuning/time_d.py  (run under rocprofv3 --kernel-trace --stats for the kernel split)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from gan_sr_wind_field_amd.CNN_models.Discriminator_3D import Discriminator_3D  # noqa: E402
from gan_sr_wind_field_amd.tools import initialization  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    torch.manual_seed(0)
    D = Discriminator_3D(3, 32, number_of_z_layers=128, enable_slicing=False, use_mixed_precision=True,
                         dropout_probability=0.2).to(DEV)
    initialization.init_weights(D, scale=0.2)
    x = (torch.rand((1, 3, 128, 128, 128), device=DEV) * 2 - 1)

    def fwd_eval():
        D.eval()
        with torch.no_grad():
            D(x)

    def fwd_bwd_input():  # G-iteration: D.eval(), frozen parameters, gradient w.r.t. the input
        D.eval()
        for p in D.parameters():
            p.requires_grad = False
        xg = x.clone().requires_grad_(True)
        D(xg).sum().backward()

    def fwd_bwd_params():  # D-iteration: train mode, parameter gradients
        D.train()
        for p in D.parameters():
            p.requires_grad = True
        D.zero_grad(set_to_none=True)
        D(x).sum().backward()

    print(f"D forward, eval, no grad            {timeit(fwd_eval):8.3f} ms")
    print(f"D forward + input gradient (eval)   {timeit(fwd_bwd_input):8.3f} ms")
    print(f"D forward + parameter grads (train) {timeit(fwd_bwd_params):8.3f} ms")


if __name__ == "__main__":
    main()
