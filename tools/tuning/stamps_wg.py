"""In-kernel phase stamps of the tile wgrad kernel (needs the -DWSR_CT_STAMPS library: WSR_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gan_sr_wind_field_amd import hip_ops as o
DEV, DT = "cuda:0", torch.bfloat16

def run(name, cin, cout, k, xyz, in_ctot, out_ctot, out_off, tri=None, ups=False):
    B = 1
    geom = o.ConvGeom(cin, cout, k, (1, 1, 1), tuple(kk // 2 for kk in k), upsample=ups)
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn((B,) + xyz + (in_ctot,), device=DEV, generator=g).to(DT)
    d = o.make_desc(geom, DT, B, xyz, in_ctot, 0, out_ctot, out_off)
    oxyz = (d.Xo, d.Yo, d.Zo)
    gy = torch.randn((B,) + oxyz + (out_ctot,), device=DEV, generator=g).to(DT)
    stamps = torch.zeros((4096, 8), dtype=torch.int64, device=DEV)
    if os.environ.get("WG_ATOMIC"):  # the first form: float atomics into one copy
        dw = torch.zeros((cout, geom.taps, cin), dtype=torch.float32, device=DEV)
        fn = (lambda: o.conv_wgrad_tri(d, x, gy, dw, *tri)) if tri else (lambda: o.conv_wgrad(d, x, gy, dw))
    else:                            # what the step runs: one split copy per workgroup, plain stores
        tb, tstep = tri if tri else (0, 0)
        n = o.conv_wgrad_nparts(d, tb, tstep)
        parts = torch.empty((n, cout, geom.taps, cin), dtype=torch.float32, device=DEV)
        fn = lambda: o.conv_wgrad_parts(d, x, gy, parts, n, tb, tstep)
        name += f" [{n} split copies, {parts.numel() * 4 / 1e6:.1f} MB]"
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    os.environ["WSR_CT_STAMPS_PTR"] = hex(stamps.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    del os.environ["WSR_CT_STAMPS_PTR"]
    s = stamps.cpu().numpy()
    s = s[s[:, 6] != 0]
    t0 = s[:, 0].min()
    us = lambda a: a / 100.0
    print(f"== {name}: event {e0.elapsed_time(e1)*1e3:.1f} us, {len(s)} WGs (running)")
    print("   WG start (us after first): mean %.1f max %.1f" % (us(s[:, 0] - t0).mean(), us(s[:, 0] - t0).max()))
    for i, n in enumerate(["prologue", "tile loop", "flush"]):
        dlt = us(s[:, i + 1] - s[:, i])
        print(f"   {n:10s} mean {dlt.mean():7.2f}  min {dlt.min():7.2f}  max {dlt.max():7.2f} us")
    clk = ((s[:, 7] - s[:, 6]) / (s[:, 3] - s[:, 0]) * 100.0).mean()
    tiles = s[:, 5].mean()
    print("   wave 0: barrier+dma wait %.0f clk per tile iteration, %.1f iterations; loop %.0f clk/iteration; clock %.0f MHz; last end %.1f us" % (
        (s[:, 4] / s[:, 5]).mean(), tiles, (us(s[:, 2] - s[:, 1]) * clk / s[:, 5]).mean(), clk, us(s[:, 3].max() - t0)))

LR = (32, 32, 128)
CASES = {
    "lr": lambda: run("lr_conv 128->128 k3", 128, 128, (3, 3, 3), LR, 128, 128, 0),
    "rdb": lambda: run("rdb stacked (4 convs) k3", 224, 128, (3, 3, 3), LR, 256, 256, 128, tri=(128, 32)),
    "up": lambda: run("up2 128->128 k3 (64^2 -> 128^2)", 128, 128, (3, 3, 3), (64, 64, 128), 128, 128, 0, ups=True),
    "hr0": lambda: run("hr0 144->144 k5", 144, 144, (5, 5, 5), (128, 128, 128), 144, 144, 0),
    "lff": lambda: run("lff 256->128 k1", 256, 128, (1, 1, 1), LR, 256, 128, 0),
    "d0": lambda: run("D conv0 3(8)->32 k3 @128^3", 8, 32, (3, 3, 3), (128, 128, 128), 8, 32, 0),
    "t1": lambda: run("terrain1 16->16 k3 @128^3", 16, 16, (3, 3, 3), (128, 128, 128), 16, 16, 0),
}
for n in (sys.argv[1:] or list(CASES)):
    CASES[n]()
