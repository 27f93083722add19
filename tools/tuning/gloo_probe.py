"""Probe: N ranks on ONE card over gloo - do concurrent asynchronous all-reduces of device tensors complete?
(torchrun --nproc-per-node N tools/tuning/gloo_probe.py; used to tell a transport hang from a bug in the step)"""
import sys
import time

import torch
import torch.distributed as dist

dist.init_process_group("gloo")
r = dist.get_rank()
dev = "cuda:0" if torch.cuda.is_available() else "cpu"
sizes = [int(a) for a in sys.argv[1:]] or [8388608] * 4 + [1400000, 500000, 200000, 16, 33]
ts = [torch.full((n,), float(r), device=dev) for n in sizes]
t0 = time.time()
g = torch.ones(6, device=dev)
dist.all_reduce(g)
ws = [dist.all_reduce(t, async_op=True) for t in ts]
for w in ws:
    w.wait()
if dev != "cpu":
    torch.cuda.synchronize()
print(r, "ok", round(time.time() - t0, 3), ts[0][0].item(), flush=True)
