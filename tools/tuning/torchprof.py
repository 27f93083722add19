"""attribute the torch-side kernels (fills, adds, copies) of one bench step to Python call sites"""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
args = argparse.Namespace(gpus=1, steps=1, warmup=1, dtype="bf16", batch=1, n=32, nz=128, no_cpu_baseline=True)
dev = torch.device("cuda:0")
gan, cfg = bench.make_gan(args, dev, "bf16")
from oracle.gan import synthetic_batch
LR, HR, Z, x, y = (t.to(dev) for t in synthetic_batch(1, 32, 128, cfg.scale, seed=2001))
gan.feed_xy_niter(x, y, torch.tensor(cfg.training.niter, device=dev), 1, 1)
def step(i):
    gan.optimize_parameters(LR, HR, Z, 2 * i)
    gan.optimize_parameters(LR, HR, Z, 2 * i + 1)
step(0); step(1)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(2)
    torch.cuda.synchronize()
import collections
cnt = collections.Counter(); tim = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::copy_", "aten::sum", "aten::mul", "aten::zeros", "aten::clone"):
        st = [f for f in (e.stack or []) if "gan_sr_wind_field_amd" in f or "bench" in f or "torch/optim" in f or "autograd" in f]
        key = (e.name, st[0].split("/")[-1][:90] if st else "?", str(e.input_shapes)[:60])
        cnt[key] += 1
        tim[key] += e.device_time_total
for k, n in cnt.most_common(45):
    print(f"{n:5d} {tim[k]/1e3:8.2f} ms  {k}")
