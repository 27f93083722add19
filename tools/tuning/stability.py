#!/usr/bin/env python
"""Longer run of the train step on the HIP path (bf16, dropout, instance noise, label schedule) on synthetic batches:
prints the loss dictionaries every 50 iterations and fails on a non-finite entry.  python tools/tuning/stability.py [its]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

sys.argv = ["bench.py"] + sys.argv[1:2]
import bench  # noqa: E402


class A:
    ini, slicing, n, nz, batch, dtype = "local", False, 32, 10, 2, "bf16"


def main():
    its = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    gan, cfg = bench.make_gan(A, dev, "bf16")
    from gan_sr_wind_field_amd.process_data import synthetic_batch
    batches = [tuple(t.to(dev) for t in synthetic_batch(A.batch, A.n, A.nz, 4, seed=100 + k)) for k in range(4)]
    LR, HR, Z, x, y = batches[0]
    gan.feed_xy_niter(x, y, torch.tensor(cfg.training.niter, device=dev), 1, 1)
    w0 = {k: v.clone() for k, v in gan.G.state_dict().items()}
    for it in range(its):
        LR, HR, Z, _, _ = batches[it % len(batches)]
        gan.optimize_parameters(LR, HR, Z, it)
        if it > 2 * cfg.training.d_g_train_period:
            gan.update_learning_rate()
        if it % 50 in (0, 1) or it >= its - 2:
            g = {k: float(v) for k, v in gan.get_G_train_loss_dict_ref().items()}
            d = float(gan.get_D_loss_dict_ref()["train_loss"])
            ok = all(v == v and abs(v) != float("inf") for v in list(g.values()) + [d])
            print(f"it {it:4d} G total {g['total']:.4f} pix {g['pix']:.4f} adv {g['adversarial']:.4f} "
                  f"div {g['divergence']:.4f} | D {d:.4f} {'ok' if ok else 'NON-FINITE'}", flush=True)
            assert ok
    moved = max(float((gan.G.state_dict()[k].float() - w0[k].float()).abs().max()) for k in w0 if w0[k].is_floating_point())
    finite = all(bool(torch.isfinite(v.float()).all()) for v in gan.G.state_dict().values())
    print(f"max |dW| of G {moved:.3e}; all weights finite: {finite}; logs: {gan.get_new_status_logs()[-3:]}")
    assert finite and moved > 0


if __name__ == "__main__":
    main()
