"""In-kernel phase stamps of a tile conv launch (needs the -DWSR_CT_STAMPS library: WSR_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gan_sr_wind_field_amd import hip_ops as o
DEV, DT = "cuda:0", torch.bfloat16

def run(name, cin, cout, k, xyz, in_ctot, out_ctot, out_off, what="fwd", red_dy=None):
    B = 1
    geom = o.ConvGeom(cin, cout, k, (1, 1, 1), tuple(kk // 2 for kk in k))
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn((B,) + xyz + (in_ctot,), device=DEV, generator=g).to(DT)
    w = torch.randn((cout, cin) + k, device=DEV, generator=g) * 0.05
    d = o.make_desc(geom, DT, B, xyz, in_ctot, 0, out_ctot, out_off)
    y = torch.zeros((B,) + xyz + (out_ctot,), dtype=DT, device=DEV)
    stamps = torch.zeros((3 * 4096 + 8, 8), dtype=torch.int64, device=DEV)
    if what == "fwd":
        wf = o.pack_filter_frag(w)
        fn = lambda: o.conv_fwd_tile(d, x, wf, y, act=True)
    else:
        wf = o.pack_filter_frag(w, transpose=True)
        fn = lambda: o.conv_dgrad_tile(d, y, wf, x, accumulate=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    os.environ["WSR_CT_STAMPS_PTR"] = hex(stamps.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    del os.environ["WSR_CT_STAMPS_PTR"]
    sall = stamps.cpu().numpy()
    nwg = int((sall[:, 6] != 0).sum())
    s = sall[:nwg]
    w = sall[nwg:3 * nwg].reshape(nwg, 2, 8)
    t0 = s[:, 0].min()
    us = lambda a: a / 100.0  # 100 MHz
    print(f"== {name} {what}: event {e0.elapsed_time(e1)*1e3:.1f} us, {len(s)} WGs")
    print("   WG start  (us after first): mean %.1f max %.1f" % (us(s[:, 0] - t0).mean(), us(s[:, 0] - t0).max()))
    names = ["w-iss+addr", "x-issue", "tabs+fill", "main-loop", "epilogue"]
    for i, n in enumerate(names):
        dlt = us(s[:, i + 1] - s[:, i])
        print(f"   {n:10s} mean {dlt.mean():6.2f}  min {dlt.min():6.2f}  max {dlt.max():6.2f} us")
    tot = us(s[:, 5] - s[:, 0])
    print("   WG total mean %.2f max %.2f; last WG end %.1f us after first start" % (tot.mean(), tot.max(), us(s[:, 5].max() - t0)))
    for k, nm in ((0, "first wave"), (1, "last wave ")):
        ph = w[:, k, 2].mean()
        print("   %s: dma_wait %.0f clk/phase, barrier %.0f clk/phase, %d phases; loop clk/phase %.0f" % (
            nm, w[:, k, 0].mean() / ph, w[:, k, 1].mean() / ph, ph, 0))
    clk = (s[:, 7] - s[:, 6]) / (s[:, 5] - s[:, 0]) * 100.0
    print("   shader clock %.0f MHz (mean)" % clk.mean())

LR = (32, 32, 128)
if len(sys.argv) > 1 and sys.argv[1] == "hr0":
    run("hr0 144->144 k5 (N=144 tile)", 144, 144, (5, 5, 5), (128, 128, 128), 144, 144, 0)
    sys.exit(0)
run("pre 128->128 (N=128 tile)", 128, 128, (3, 3, 3), LR, 256, 256, 128)
run("grow 64->32", 64, 32, (3, 3, 3), LR, 256, 256, 192)
run("up2 128->128 @128^2", 128, 128, (3, 3, 3), (128, 128, 128), 128, 128, 0)
