#!/usr/bin/env python
"""Layer-by-layer distance of the bf16 discriminator program from the fp32 oracle, next to the oracle's own
bf16-storage emulation (diagnostic for the bf16 tolerances; test infrastructure, GPU box only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import nets as onets  # noqa: E402
from test_hip_networks import build_D  # noqa: E402


def rl(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def oracle_layers(sd, x, s, training):
    outs = []
    x = onets._st(x, s)
    for l in onets.d_layers(s):
        x = F.conv3d(x, onets._wq(sd[l.key + ".weight"], s), None, l.stride, l.pad)
        y = None
        if l.bn:
            x = onets._st(x, s)
            y = x
            x = F.batch_norm(x, sd[l.bn + ".running_mean"].clone(), sd[l.bn + ".running_var"].clone(), sd[l.bn + ".weight"],
                             sd[l.bn + ".bias"], training, s.bn_momentum, s.bn_eps)
        if l.act:
            x = onets._lrelu(x, s.slope)
        x = onets._st(x, s)
        outs.append((y, x))
    return outs


def main():
    bf, nz, xy, slicing, seed = 8, 4, 64, True, 8
    if len(sys.argv) > 1 and sys.argv[1] == "full":
        bf, nz, xy, slicing, seed = 32, 10, 64, True, 103
    spec = onets.DSpec(bf=bf, nz=nz, enable_slicing=slicing)
    D, _ = build_D(spec, torch.bfloat16, seed)
    gen = torch.Generator().manual_seed(3)
    x = torch.rand((2, 3, xy, xy, nz), generator=gen) * 2 - 1
    D.train()
    prog = D.features.program()
    feat, saved = prog.forward(x.to("cuda:0"), True, True)
    res = {}
    for mode in (False, True):
        sd = onets.deterministic_state(onets.d_param_shapes(spec), seed=seed, scale=1.0)
        sp = onets.DSpec(bf=bf, nz=nz, enable_slicing=slicing, bf16_storage=mode)
        with torch.no_grad():
            res[mode] = oracle_layers(sd, x, sp, True)
    print(f"{'layer':16s} {'hip y':>10s} {'emul y':>10s} {'hip a':>10s} {'emul a':>10s}   mean|mu|/sigma of y")
    for l, rec, (y32, a32), (ye, ae) in zip(onets.d_layers(spec), saved["recs"], res[False], res[True]):
        c = l.cout
        hy = rec["y"][..., :c].permute(0, 4, 1, 2, 3).float() if rec["y"] is not None else None
        ha = rec["a"][..., :c].permute(0, 4, 1, 2, 3).float()
        ratio = ""
        if y32 is not None:
            mu = y32.mean(dim=(0, 2, 3, 4))
            sg = y32.std(dim=(0, 2, 3, 4))
            ratio = f"{float((mu.abs() / sg).mean()):.2f}  n={y32[:, 0].numel()}"
        print(f"{l.key:16s} {rl(hy, y32) if hy is not None else float('nan'):10.2e} "
              f"{rl(ye, y32) if ye is not None else float('nan'):10.2e} {rl(ha, a32):10.2e} {rl(ae, a32):10.2e}   {ratio}")


if __name__ == "__main__":
    main()
