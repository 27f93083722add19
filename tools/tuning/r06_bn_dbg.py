import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gan_sr_wind_field_amd import hip_ops as o
dev = "cuda:0"
for dt in (torch.float32, torch.bfloat16):
    for C, nv, G in ((8, 8192, 2), (16, 8192, 2), (64, 4096, 2), (16, 8192, 1), (256, 320, 2)):
        g = torch.Generator(device=dev).manual_seed(C + G)
        x = (torch.randn((G * 2, nv // 2, 1, 1, C), device=dev, generator=g) * 1.3 + 0.4).to(dt)
        rm0, rv0 = torch.randn(C, device=dev, generator=g), torch.rand(C, device=dev, generator=g) + 0.5
        work = torch.empty((G, 2 * C), dtype=torch.float32, device=dev)
        rm, rv = rm0.clone(), rv0.clone()
        assert o.bn_train_stats(x, G, work, 1e-5, 0.1, rm, rv)
        work2 = torch.empty((G, 2 * C), dtype=torch.float32, device=dev)
        st = torch.empty((G, 4 * C), dtype=torch.float32, device=dev)
        rm2, rv2 = rm0.clone(), rv0.clone()
        for gi in range(G):
            xg = x[gi * 2:(gi + 1) * 2]
            o.bn_stats(xg, st[gi, :2 * C])
            o.bn_mean(st[gi, :2 * C], work2[gi, :C], float(nv), None)
            o.bn_stats(xg, st[gi, 2 * C:4 * C], shift=work2[gi, :C])
            o.bn_finalize(st[gi, 2 * C:4 * C], work2[gi, :C], work2[gi, C:], float(nv), 1e-5, 0.1, rm2, rv2, None)
        torch.cuda.synchronize()
        print(dt, C, nv, G, "mean", float((work[:, :C] - work2[:, :C]).abs().max()), "invstd", float((work[:, C:] - work2[:, C:]).abs().max()),
              "rm", float((rm - rm2).abs().max()), "rv", float((rv - rv2).abs().max()))
