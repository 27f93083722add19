#!/usr/bin/env python
"""Micro-benchmark of single conv launches through the C-ABI (device time via HIP events).

    python tools/bench_conv.py [case ...]      cases: rdb hr0 hr0w rdbw up lff dg224 hr1 all
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gan_sr_wind_field_amd import hip_ops as o  # noqa: E402

DEV = "cuda:0"
DT = torch.bfloat16


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def conv_case(name, cin, cout, k, xyz, in_ctot=None, out_ctot=None, out_off=0, ups=False, what="fwd", B=1, nbytes=None):
    in_ctot = in_ctot or cin
    geom = o.ConvGeom(cin, cout, k, (1, 1, 1), tuple(kk // 2 for kk in k), upsample=ups)
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn((B,) + xyz + (in_ctot,), device=DEV, generator=g).to(DT)
    w = torch.randn((cout, cin) + k, device=DEV, generator=g) * 0.05
    d = o.make_desc(geom, DT, B, xyz, in_ctot, 0, out_ctot or cout, out_off)
    oxyz = (d.Xo, d.Yo, d.Zo)
    y = torch.zeros((B,) + oxyz + (out_ctot or cout,), dtype=DT, device=DEV)
    vox = oxyz[0] * oxyz[1] * oxyz[2]
    flops = 2.0 * vox * cin * cout * k[0] * k[1] * k[2]
    if what == "fwd":
        wf = o.pack_filter_frag(w)
        ms = timeit(lambda: o.conv_fwd_tile(d, x, wf, y, act=True))
    elif what == "fwd_generic":
        wp = o.pack_filter(w, DT)
        ms = timeit(lambda: o.conv_fwd(d, x, wp, y, act=True))
    elif what == "dgrad":
        wft = o.pack_filter_frag(w, transpose=True)
        gy = torch.randn_like(y)
        dx = torch.zeros((B,) + tuple(oxyz) + (in_ctot,), dtype=DT, device=DEV)
        ms = timeit(lambda: o.conv_dgrad_tile(d, gy, wft, dx))
    elif what == "wgrad":
        gy = torch.randn_like(y)
        dw = torch.zeros((cout, geom.taps, cin), dtype=torch.float32, device=DEV)
        ms = timeit(lambda: o.conv_wgrad(d, x, gy, dw))
    extra = f"  {nbytes / ms / 1e6:8.1f} GB/s algorithmic = {nbytes / ms / 1e6 / 8000:.3f} of 8 TB/s" if nbytes else ""
    print(f"{name:28s} {what:12s} {ms * 1e3:9.1f} us  {flops / ms / 1e9:8.1f} TF/s{extra}")


def tri_case():
    B, xyz, nf, gc, nconv = 1, (32, 32, 128), 128, 32, 4
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn((B,) + xyz + (256,), device=DEV, generator=g).to(DT)
    gd = torch.randn((B,) + xyz + (256,), device=DEV, generator=g).to(DT)
    cin_w, cout = nf + (nconv - 1) * gc, nconv * gc
    d = o.make_desc(o.ConvGeom(cin_w, cout, (3, 3, 3)), DT, B, xyz, 256, 0, 256, nf)
    dw = torch.zeros((cout, 27, cin_w), dtype=torch.float32, device=DEV)
    ms = timeit(lambda: o.conv_wgrad_tri(d, x, gd, dw, nf, gc))
    flops = 2.0 * xyz[0] * xyz[1] * xyz[2] * 27 * gc * sum(nf + i * gc for i in range(nconv))
    print(f"{'rdb stacked wgrad (4 convs)':28s} {'wgrad_tri':12s} {ms * 1e3:9.1f} us  {flops / ms / 1e9:8.1f} TF/s")


LR = (32, 32, 128)
HR = (128, 128, 128)
CASES = {
    "rdb": lambda: [conv_case(f"rdb conv{i} {128 + 32 * i}->32", 128 + 32 * i, 32, (3, 3, 3), LR, 256, 256, 128 + 32 * i)
                    for i in range(4)],
    "grow": lambda: [conv_case(f"grow stage {32 * i}->32", 32 * i, 32, (3, 3, 3), LR, 256, 256, 128 + 32 * i) for i in (1, 2, 3)]
    + [conv_case("pre 128->128", 128, 128, (3, 3, 3), LR, 256, 256, 128)],
    "rdbg": lambda: conv_case("rdb conv3 224->32", 224, 32, (3, 3, 3), LR, 256, 256, 224, what="fwd_generic"),
    "hr0": lambda: [conv_case("hr0 144->144 k5", 144, 144, (5, 5, 5), HR, what=w) for w in ("fwd", "dgrad")],
    "rdbw": tri_case,
    "hr0w": lambda: conv_case("hr0 144->144 k5", 144, 144, (5, 5, 5), HR, what="wgrad"),
    "up": lambda: [conv_case("up1 128->128 (64^2)", 128, 128, (3, 3, 3), (32, 32, 128), ups=True),
                   conv_case("up2 128->128 (128^2)", 128, 128, (3, 3, 3), (64, 64, 128), ups=True),
                   conv_case("up2 dgrad", 128, 128, (3, 3, 3), (128, 128, 128), what="dgrad"),
                   conv_case("up2 wgrad", 128, 128, (3, 3, 3), (64, 64, 128), ups=True, what="wgrad")],
    "lff": lambda: [conv_case("lff 256->128 k1", 256, 128, (1, 1, 1), LR),
                    conv_case("lff dgrad", 256, 128, (1, 1, 1), LR, what="dgrad"),
                    conv_case("lff wgrad", 256, 128, (1, 1, 1), LR, what="wgrad")],
    "dg": lambda: [conv_case(f"rdb dgrad 32->{128 + 32 * i}", 128 + 32 * i, 32, (3, 3, 3), LR, 256, 256, 0, what="dgrad")
                   for i in range(4)],
    "hr1": lambda: [conv_case("hr1 144->3 k5", 144, 3, (5, 5, 5), HR, 144, 8, what=w) for w in ("fwd", "dgrad", "wgrad")],
    "lr": lambda: [conv_case("lr_conv 128->128", 128, 128, (3, 3, 3), LR, what=w) for w in ("fwd", "dgrad", "wgrad")],
}

# the memory-bound 3x3x3 convs with a thin input side (conv_thin.hip), benchmark shapes; bytes = (Cin V + Cout V) * 2
VH, VL = 128 ** 3, 32 * 32 * 128
CASES["thin"] = lambda: [
    conv_case("terrain0 1->16", 8, 16, (3, 3, 3), HR, 8, 16, 0, nbytes=VH * 17 * 2),
    conv_case("terrain1 16->16 (concat)", 16, 16, (3, 3, 3), HR, 16, 144, 128, nbytes=VH * 32 * 2),
    conv_case("terrain1 dgrad", 16, 16, (3, 3, 3), HR, 16, 16, 0, what="dgrad", nbytes=VH * 32 * 2),
    conv_case("feature 4->128", 8, 128, (3, 3, 3), LR, 8, 256, 0, nbytes=VL * 132 * 2),
    conv_case("D first 3->32 (pair)", 8, 32, (3, 3, 3), HR, 8, 32, 0, B=2, nbytes=2 * VH * 35 * 2),
]


def up_parity_wgrad_case():
    B, xyz, c = 1, (64, 64, 128), 128
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn((B,) + xyz + (c,), device=DEV, generator=g).to(DT)
    gy = torch.randn((B, 128, 128, 128, c + 16), device=DEV, generator=g).to(DT)
    tot = 0.0
    for ph in range(4):
        a, b = ph >> 1, ph & 1
        d = o.make_desc(o.ConvGeom(c, c, (2, 2, 3), (1, 1, 1), (1 - a, 1 - b, 1)), DT, B, xyz, c, 0, c + 16, 0, lat=(a, b, 0))
        n = o.conv_wgrad_nparts(d)
        parts = torch.empty((n, c, 12, c), dtype=torch.float32, device=DEV)
        tot += timeit(lambda: o.conv_wgrad_parts(d, x, gy, parts, n))
    flops = 2.0 * 4 * xyz[0] * xyz[1] * xyz[2] * 12 * c * c
    print(f"{'up2 wgrad, 4 parity launches':28s} {'wgrad':12s} {tot * 1e3:9.1f} us  {flops / tot / 1e9:8.1f} TF/s")


CASES["upw"] = up_parity_wgrad_case

S10 = (32, 32, 10)  # the reference's real patch size at the trunk's resolution (C1b)
CASES["c1b"] = lambda: [conv_case("pre 128->128 @32x32x10", 128, 128, (3, 3, 3), S10, 256, 256, 128),
                        conv_case("pre dgrad @32x32x10", 128, 128, (3, 3, 3), S10, 256, 256, 0, what="dgrad"),
                        conv_case("grow 96->32 @32x32x10", 96, 32, (3, 3, 3), S10, 256, 256, 224),
                        conv_case("up2 128->128 @64x64x10", 128, 128, (3, 3, 3), (64, 64, 10), ups=True),
                        conv_case("hr0 @128x128x10", 144, 144, (5, 5, 5), (128, 128, 10)),
                        conv_case("lr wgrad @32x32x10", 128, 128, (3, 3, 3), S10, what="wgrad")]

def hr1z_case():
    """the z-folded last conv at the benchmark's HR shape: forward 144 -> 15 (planar fp32 out) and its input
    gradient 16 -> 144 with the LeakyReLU / Dropout3d mask of hr_convs[0] in the epilogue"""
    B, xyz, c, n = 1, HR, 144, 15
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn((B,) + xyz + (c,), device=DEV, generator=g).to(DT)
    w = torch.randn((n, c, 5, 5, 1), device=DEV, generator=g) * 0.02
    vox = xyz[0] * xyz[1] * xyz[2]
    d = o.make_desc(o.ConvGeom(c, n, (5, 5, 1), (1, 1, 1), (2, 2, 0)), DT, B, xyz, c, 0, n, 0)
    y = torch.empty((B, n) + xyz, dtype=torch.float32, device=DEV)
    wf = o.pack_filter_frag(w)
    ms = timeit(lambda: o.conv_fwd_tile(d, x, wf, y, out_planar=True))
    nb = vox * (c * 2 + n * 4)
    print(f"{'hr1z fwd 144->15 (5,5,1)':28s} {'fwd planar':12s} {ms * 1e3:9.1f} us  {nb / ms / 1e6:8.1f} GB/s algorithmic")
    gy = torch.randn((B,) + xyz + (16,), device=DEV, generator=g).to(DT)
    gy[..., 15] = 0
    dx = torch.empty((B,) + xyz + (c,), dtype=DT, device=DEV)
    drop = torch.ones((B, c), device=DEV)
    dd = o.make_desc(o.ConvGeom(c, 16, (5, 5, 1), (1, 1, 1), (2, 2, 0)), DT, B, xyz, c, 0, 16, 0)
    wft = o.pack_filter_frag(torch.cat([w, torch.zeros_like(w[:1])]), transpose=True)
    ms = timeit(lambda: o.conv_dgrad_tile(dd, gy, wft, dx, mask=(x, 0, 0, c, 0.2, drop)))
    nb = vox * (16 * 2 + 2 * c * 2)
    print(f"{'hr1z dgrad 16->144 +mask':28s} {'dgrad':12s} {ms * 1e3:9.1f} us  {nb / ms / 1e6:8.1f} GB/s algorithmic")
    # filter gradient, as stated (x the halo image, one n-tile) and with the operands' roles exchanged (engine.py
    # SWAP_THIN_WGRAD: dy the halo image, 48 of x's channels per workgroup); the two must agree after the tap flip
    dw = o.make_desc(o.ConvGeom(c, 16, (5, 5, 1), (1, 1, 1), (2, 2, 0)), DT, B, xyz, c, 0, 16, 0)
    n1 = o.conv_wgrad_nparts(dw)
    p1 = torch.empty((n1, 16, 25, c), dtype=torch.float32, device=DEV)
    ms1 = timeit(lambda: o.conv_wgrad_parts(dw, x, gy, p1, n1))
    ds = o.make_desc(o.ConvGeom(16, c, (5, 5, 1), (1, 1, 1), (2, 2, 0)), DT, B, xyz, 16, 0, c, 0)
    n2 = o.conv_wgrad_nparts(ds)
    p2 = torch.empty((n2, c, 25, 16), dtype=torch.float32, device=DEV)
    ms2 = timeit(lambda: o.conv_wgrad_parts(ds, gy, x, p2, n2))
    nb = vox * (16 * 2 + c * 2)
    print(f"{'hr1z wgrad as stated':28s} {'wgrad':12s} {ms1 * 1e3:9.1f} us  {nb / ms1 / 1e6:8.1f} GB/s algorithmic ({n1} copies)")
    print(f"{'hr1z wgrad roles exchanged':28s} {'wgrad':12s} {ms2 * 1e3:9.1f} us  {nb / ms2 / 1e6:8.1f} GB/s algorithmic ({n2} copies)")
    g1 = p1.double().sum(0)                                   # [n][tap][c]
    g2 = p2.double().sum(0).flip(1).permute(2, 1, 0)          # [c][K-1-tap][n] -> [n][tap][c]
    print(f"   exchanged vs stated: rel-L2 {float((g1 - g2).norm() / g1.norm()):.2e}")


CASES["hr1z"] = hr1z_case

if __name__ == "__main__":
    names = sys.argv[1:] or ["all"]
    if names == ["all"]:
        names = list(CASES)
    for n in names:
        CASES[n]()
