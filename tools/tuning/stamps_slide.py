"""Per-phase cycle sums of the sliding-window forward kernel (needs the -DWSR_CS_STAMPS library: WSR_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gan_sr_wind_field_amd import hip_ops as o
DEV, DT = "cuda:0", torch.bfloat16
B, xyz, c, n = 1, (128, 128, 128), 144, 15
g = torch.Generator(device=DEV).manual_seed(1)
x = torch.randn((B,) + xyz + (c,), device=DEV, generator=g).to(DT)
w = torch.randn((n, c, 5, 5, 1), device=DEV, generator=g) * 0.02
d = o.make_desc(o.ConvGeom(c, n, (5, 5, 1), (1, 1, 1), (2, 2, 0)), DT, B, xyz, c, 0, n, 0)
y = torch.empty((B, n) + xyz, dtype=torch.float32, device=DEV)
wf = o.pack_filter_frag(w)
fn = lambda: o.conv_fwd_tile(d, x, wf, y, out_planar=True)
for _ in range(3):
    fn()
torch.cuda.synchronize()
stamps = torch.zeros((256 * 8, 8), dtype=torch.int64, device=DEV)
os.environ["WSR_CS_STAMPS_PTR"] = hex(stamps.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record()
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(256, 8, 8).astype(float)
print(f"event {e0.elapsed_time(e1) * 1e3:.1f} us; planes {s[0, 0, 7]:.0f}")
if os.environ.get("WSR_CS_WALL"):  # library built with WSR_CS_STAMPS_MODE=2: wall-clock (100 MHz) stamps around the loop
    w0 = s[:, :, 0].min()
    us = lambda a: (a - w0) / 100.0
    print("kernel entry  (us after the first wave): mean %.1f max %.1f" % (us(s[:, :, 0]).mean(), us(s[:, :, 0]).max()))
    print("loop start    mean %.1f  min %.1f max %.1f" % (us(s[:, :, 1]).mean(), us(s[:, :, 1]).min(), us(s[:, :, 1]).max()))
    print("loop end      mean %.1f  min %.1f max %.1f" % (us(s[:, :, 2]).mean(), us(s[:, :, 2]).min(), us(s[:, :, 2]).max()))
    print("kernel end    mean %.1f  min %.1f max %.1f" % (us(s[:, :, 3]).mean(), us(s[:, :, 3]).min(), us(s[:, :, 3]).max()))
    print("loop clocks per plane %.0f -> shader clock %.0f MHz" % ((s[:, :, 6] / s[:, :, 7]).mean(), (s[:, :, 6] / (s[:, :, 2] - s[:, :, 1]) * 100).mean()))
    sys.exit(0)
names = ["dma issue", "reads+mfma", "finalize", "partial wr", "vmcnt wait", "barrier"]
for wv in range(8):
    per = s[:, wv, :6].mean(axis=0) / s[:, wv, 7].mean()
    tot = s[:, wv, 6].mean() / s[:, wv, 7].mean()
    print(f"wave {wv} (kg {wv & 3}, mh {wv >> 2}): " + "  ".join(f"{nm} {v:6.0f}" for nm, v in zip(names, per)) + f"  | loop {tot:6.0f} clk/plane")
