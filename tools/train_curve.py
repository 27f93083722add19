#!/usr/bin/env python
"""bf16 against fp32 over a few hundred iterations of the shipped configuration (does bf16 storage train?).

Both runs start from the same seed (same initial G / D, same synthetic batches, the same label / noise / dropout
draws as long as the branches agree) and run ``wind_field_GAN_3D.optimize_parameters`` for ``--its`` iterations of
the C1 shipped configuration (``wind_field_GAN_3D_config_local.ini``: LR 16x16x10 -> HR 64x64x10, D with slicing,
dropout, instance noise, noisy labels, both Adam steps; reference GAN_models/wind_field_GAN_3D.py:532-568 is the
schedule that alternates D- and G-iterations).  Every iteration's 8 generator loss entries and the discriminator
loss are recorded; the JSON holds the two curves and a summary of their distance:

    python tools/train_curve.py --its 300 --out profiles/r03_bf16_vs_fp32_curve.json

The synthetic dataset is 8 rotating batches of 2 samples (no network for HARMONIE-SIMRA files); the question asked
is numerical - do the bf16 gradients (30 % relative error per D tensor at batch 1) drive the same
optimisation as the reference's fp32 - not whether the GAN converges on real data.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

_argv, sys.argv = sys.argv, ["bench.py"]
import bench  # noqa: E402
sys.argv = _argv


class Shape:
    ini, slicing, n, nz, batch = "local", True, 16, 10, 2


def run(dtype, its, d_g_period, seed=2001):
    from gan_sr_wind_field_amd.process_data import synthetic_batch

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.manual_seed(seed)
    gan, cfg = bench.make_gan(Shape, dev, dtype)
    cfg.training.d_g_train_period = d_g_period
    batches = [tuple(t.to(dev) for t in synthetic_batch(Shape.batch, Shape.n, Shape.nz, 4, seed=500 + k)) for k in range(8)]
    x, y = batches[0][3], batches[0][4]
    gan.feed_xy_niter(x, y, torch.tensor(cfg.training.niter, device=dev), 1, 1)
    torch.manual_seed(seed + 1)  # the step's own draws (dropout masks, instance noise, label noise)
    curve = {"G": {}, "D": []}
    for it in range(its):
        LR, HR, Z, _, _ = batches[it % len(batches)]
        gan.optimize_parameters(LR, HR, Z, it)
        gan.update_learning_rate()
        g = {k: float(v) for k, v in gan.get_G_train_loss_dict_ref().items()}
        for k, v in g.items():
            curve["G"].setdefault(k, []).append(v)
        curve["D"].append(float(gan.get_D_loss_dict_ref()["train_loss"]))
    finite = all(bool(torch.isfinite(v.float()).all()) for v in gan.G.state_dict().values())
    # validation-style figure of merit on a held-out synthetic batch: pixel L1 of G(LR) against HR, eval mode
    LRv, HRv, Zv, _, _ = (t.to(dev) for t in synthetic_batch(2, Shape.n, Shape.nz, 4, seed=9001))
    gan.G.eval()
    with torch.no_grad():
        pix = float((gan.G(LRv, Zv) - HRv).abs().mean())
    gan.G.train()
    return curve, finite, pix


def tail_mean(v, n=50):
    return sum(v[-n:]) / len(v[-n:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--its", type=int, default=300)
    ap.add_argument("--d-g-period", type=int, default=1, help="iterations per D / G phase (1: alternate every iteration)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bf16_vs_fp32_curve.json"))
    args = ap.parse_args()
    res = {}
    for dt in ("fp32", "bf16"):
        curve, finite, pix = run(dt, args.its, args.d_g_period)
        res[dt] = {"curve": curve, "all_weights_finite": finite, "heldout_pix_l1": pix}
        print(f"{dt}: finite {finite}, held-out pixel L1 {pix:.5f}, last-50 mean G total "
              f"{tail_mean(curve['G']['total']):.5f}, D {tail_mean(curve['D']):.5f}", flush=True)
    f32, b16 = res["fp32"]["curve"], res["bf16"]["curve"]
    summ = {}
    for k in list(f32["G"]) + ["D"]:
        a = f32["G"][k] if k != "D" else f32["D"]
        b = b16["G"][k] if k != "D" else b16["D"]
        # entries only change on their own kind of iteration; compare the windows' means, not single draws
        first, last = slice(0, 50), slice(-50, None)
        ma0, mb0 = sum(a[first]) / 50, sum(b[first]) / 50
        ma1, mb1 = tail_mean(a), tail_mean(b)
        summ[k] = {"fp32_first50": ma0, "bf16_first50": mb0, "fp32_last50": ma1, "bf16_last50": mb1,
                   "rel_gap_last50": abs(mb1 - ma1) / max(abs(ma1), 1e-12),
                   "max_abs_gap": max(abs(p - q) for p, q in zip(a, b))}
    res["summary"] = summ
    res["config"] = {"its": args.its, "d_g_train_period": args.d_g_period, "shape": "C1 shipped ini, LR 16x16x10 -> HR 64x64x10, "
                     "batch 2, D sliced, 8 rotating synthetic batches", "seed": 2001}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f)
    for k, v in summ.items():
        print(f"{k:14s} fp32 {v['fp32_first50']:.5f} -> {v['fp32_last50']:.5f} | bf16 {v['bf16_first50']:.5f} -> "
              f"{v['bf16_last50']:.5f} | rel gap of the last-50 means {v['rel_gap_last50']:.3f}")
    assert res["fp32"]["all_weights_finite"] and res["bf16"]["all_weights_finite"]


if __name__ == "__main__":
    main()
