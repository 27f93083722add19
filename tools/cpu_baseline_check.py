#!/usr/bin/env python
"""Is the CPU baseline honest?  Times, in the BUILD container (where /root/reference exists) and at the same
thread count, (a) the REAL reference - ``wind_field_GAN_3D.optimize_parameters`` of jacobwulffwold/GAN_SR_wind_field -
and (b) the oracle that ``bench.py``'s ``cpu_baseline`` leg runs on the GPU box, on the same workload: the shipped
local configuration at full width (C1: LR 16x16x10 -> HR 64x64x10, batch 1), G-iteration + D-iteration pairs with
dropout and instance noise ON as shipped.  BASELINE.md section 4 asks for the restatement to land within +-20 % of
the reference.

    python tools/cpu_baseline_check.py [threads] [C1|C1b] > profiles/r03_cpu_baseline_check.txt

``C1b`` is the reference's real patch size (LR 32x32x10 -> HR 128x128x10, D without slicing) - four times
the generator work of C1 per pair; three timed pairs instead of nine.
"""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)
import torch  # noqa: E402


def one_pair(step, i):
    t0 = time.time()
    step(2 * i), step(2 * i + 1)
    return time.time() - t0


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    shape = sys.argv[2] if len(sys.argv) > 2 else "C1"
    n_lr, slicing, reps = {"C1": (16, True, 9), "C1b": (32, False, 3)}[shape]
    torch.set_num_threads(threads)
    from oracle import gan as ogan
    from oracle import nets as onets

    LR, HR, Z, x, y = ogan.synthetic_batch(1, n_lr, 10, 4, seed=2001)
    # ---- (a) the reference itself
    nc = types.ModuleType("netCDF4")
    nc.Dataset = nc.MFDataset = object
    sys.modules.setdefault("netCDF4", nc)
    from GAN_models import wind_field_GAN_3D as ref_gan
    from config.config import Config as RefConfig

    cfg = RefConfig(os.path.join(REF, "config", "wind_field_GAN_3D_config_local.ini"))
    cfg.is_train, cfg.is_test, cfg.is_use = True, False, False
    cfg.gpu_id, cfg.device = None, torch.device("cpu")
    cfg.training.niter, cfg.training.d_g_train_period = 150000, 1
    cfg.gan_config.enable_slicing = slicing
    torch.manual_seed(2001)
    gan = ref_gan.wind_field_GAN_3D(cfg)
    gan.feed_xy_niter(x, y, torch.tensor(cfg.training.niter), 1, 1)
    ref_step = lambda it: gan.optimize_parameters(LR, HR, Z, it)  # noqa: E731
    # ---- (b) the oracle, exactly as bench.py's cpu_baseline builds it
    gs = onets.GSpec(dropout_p=0.1)
    ds = onets.DSpec(bf=32, nz=10, enable_slicing=slicing, dropout_p=0.2)
    gen = torch.Generator().manual_seed(0)
    sdG, sdD = onets.make_state(onets.g_param_shapes(gs)), onets.make_state(onets.d_param_shapes(ds))
    onets.kaiming_init_(sdG, 0.1, gen)
    onets.kaiming_init_(sdD, 0.2, gen)
    og = ogan.OracleGAN(sdG, sdD, gs, ds, ogan.TrainSpec(d_g_train_period=1))
    og.feed_xy(x, y)
    or_step = lambda it: og.optimize_parameters(LR, HR, Z, it)  # noqa: E731
    # interleaved (the shared host drifts by tens of per cent over a minute), one warm-up pair each, median of 5
    one_pair(ref_step, 0), one_pair(or_step, 0)
    tr, to = [], []
    for i in range(1, reps + 1):
        tr.append(one_pair(ref_step, i))
        to.append(one_pair(or_step, i))
    t_ref, t_or = sorted(tr)[reps // 2], sorted(to)[reps // 2]
    ratio = t_or / t_ref
    print(f"shape {shape}: LR {n_lr}x{n_lr}x10 -> HR {4 * n_lr}x{4 * n_lr}x10, batch 1, D {'sliced' if slicing else 'full'}")
    print(f"threads {threads}  torch {torch.__version__}  host cores {os.cpu_count()}")
    print(f"reference  wind_field_GAN_3D.optimize_parameters  G-it + D-it pair: median {t_ref:.3f} s  "
          f"(runs {' '.join(f'{v:.2f}' for v in tr)})")
    print(f"oracle     oracle.gan.OracleGAN.optimize_parameters G-it + D-it pair: median {t_or:.3f} s  "
          f"(runs {' '.join(f'{v:.2f}' for v in to)})")
    print(f"oracle / reference = {ratio:.3f}   (BASELINE.md 4: within 0.8 .. 1.2: {'OK' if 0.8 <= ratio <= 1.2 else 'OUTSIDE'})")
    return 0 if 0.8 <= ratio <= 1.2 else 1


if __name__ == "__main__":
    sys.exit(main())
