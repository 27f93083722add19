#!/usr/bin/env python
"""Host-side cost of issuing one train step: cProfile of the step loop (GPU box).  python tools/tuning/hostprof2.py"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

sys.argv = ["bench.py"]
import bench  # noqa: E402


class A:
    ini, slicing, n, nz, batch, dtype = "local", False, 32, int(os.environ.get("HP_NZ", "128")), 1, "bf16"


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    gan, cfg = bench.make_gan(A, dev, "bf16")
    from gan_sr_wind_field_amd.process_data import synthetic_batch
    LR, HR, Z, x, y = (t.to(dev) for t in synthetic_batch(1, 32, A.nz, 4))
    gan.feed_xy_niter(x, y, torch.tensor(150000, device=dev), 1, 1)

    def step(i):
        gan.optimize_parameters(LR, HR, Z, 2 * i)
        gan.optimize_parameters(LR, HR, Z, 2 * i + 1)

    for i in range(2):
        step(i)
    torch.cuda.synchronize()
    # host time alone: issue 3 steps, note when the host is done and when the GPU is
    t0 = time.perf_counter()
    for i in range(2, 5):
        step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host issue {1e3 * (t1 - t0) / 3:.1f} ms/step, wall {1e3 * (t2 - t0) / 3:.1f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for i in range(5, 8):
        step(i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
