#!/usr/bin/env python
"""GAN train-steps/sec of the MI355X-native wind-field GAN (BASELINE.json metric).

A step = one G-iteration + one D-iteration (reference wind_field_GAN_3D.py:585-593
alternates the two kinds) on one synthetic batch already resident in HBM.
Workload at N = 1: BASELINE.json configs[2] in reference semantics (SURVEY 8d "C3'"):
LR (B,4,32,32,128) -> HR (B,3,128,128,128), full-size G (16 RRDB, nf 128) and the
128^3 D, bf16 compute with fp32 loss, dropout / instance noise / Adam all on.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|fp32] [--batch B]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see the keys below).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# per-sample algorithmic FLOPs, SURVEY.md 8(d) formulae
def g_fwd_flops(n, nz, s=4, nf=128, gc=32, tf=16, cin=4, nrrdb=16):
    v = n * n * nz
    V = (s * n) ** 2 * nz
    rdb = sum(27 * (nf + i * gc) * gc for i in range(4)) + (nf + 4 * gc) * nf
    trunk = v * (27 * cin * nf + nrrdb * 3 * rdb + 27 * nf * nf)
    ups, m = 0, v
    k = 0
    while (1 << k) < s:
        m *= 4
        ups += m * 27 * nf * nf
        k += 1
    c = nf + tf
    hr = V * (27 * tf + 27 * tf * tf + 125 * c * c + 125 * c * 3)
    return 2 * (trunk + ups + hr)


def d_fwd_flops(xy, nz, bf=32):
    """no-slicing D on (3, xy, xy, nz); z halves in block 0 when nz > 19 and in block 4"""
    f, X, Z = 0, xy, nz
    cin = 3
    for i, cout in enumerate((bf, 2 * bf, 4 * bf, 8 * bf, 8 * bf)):
        f += 2 * cin * cout * 27 * X * X * Z
        halve = (i == 0 and nz > 19) or i == 4
        Zo = (Z + 2 - 3) // 2 + 1 if halve else Z
        X //= 2
        f += 2 * cout * cout * 48 * X * X * Zo
        Z, cin = Zo, cout
    return f + 2 * (8 * bf * 16 * Z) * 100


def make_gan(args, dev, dtype):
    from gan_sr_wind_field_amd.config.config import Config
    from gan_sr_wind_field_amd.GAN_models.wind_field_GAN_3D import wind_field_GAN_3D

    cfg = Config(os.path.join(ROOT, "gan_sr_wind_field_amd", "config", "wind_field_GAN_3D_config_local.ini"))
    cfg.is_train, cfg.is_test, cfg.is_use = True, False, False
    cfg.gpu_id, cfg.device = dev.index, dev
    cfg.compute_dtype = dtype
    cfg.gan_config.enable_slicing = False
    cfg.gan_config.number_of_z_layers = args.nz
    cfg.training.niter = 150000
    cfg.training.d_g_train_period = 1  # it even -> G-iteration, it odd -> D-iteration
    torch.manual_seed(cfg.env.fixed_seed)
    return wind_field_GAN_3D(cfg), cfg


def cpu_baseline(n_threads):
    """The oracle (CPU restatement of the reference, kind "port") timed on this host:
    full-size G + D at the reference's own CPU-runnable case (16x16x10 -> 64x64x10, B=1),
    1 warm-up pair + timed pairs for ~15 s, converted to the metric's unit by FLOPs."""
    from oracle import gan as ogan
    from oracle import nets as onets

    torch.set_num_threads(n_threads)
    gs = onets.GSpec(dropout_p=0.1)
    ds = onets.DSpec(bf=32, nz=10, enable_slicing=True, dropout_p=0.2)
    gen = torch.Generator().manual_seed(0)
    sdG, sdD = onets.make_state(onets.g_param_shapes(gs)), onets.make_state(onets.d_param_shapes(ds))
    onets.kaiming_init_(sdG, 0.1, gen)
    onets.kaiming_init_(sdD, 0.2, gen)
    gan = ogan.OracleGAN(sdG, sdD, gs, ds, ogan.TrainSpec(d_g_train_period=1))
    LR, HR, Z, x, y = ogan.synthetic_batch(1, 16, 10, 4, seed=2001)
    gan.feed_xy(x, y)
    gan.optimize_parameters(LR, HR, Z, 0)
    gan.optimize_parameters(LR, HR, Z, 1)
    t0, pairs = time.time(), 0
    while pairs < 2 or (time.time() - t0 < 12 and pairs < 8):
        gan.optimize_parameters(LR, HR, Z, 2 * pairs + 2)
        gan.optimize_parameters(LR, HR, Z, 2 * pairs + 3)
        pairs += 1
    dt = (time.time() - t0) / pairs
    # D with slicing at 64x64x10: use the measured-table value of SURVEY 8a (8.5 GF fwd)
    pair_flops = 4 * g_fwd_flops(16, 10) + 9 * 8.5e9
    return dt, pair_flops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=1, help="samples per GPU")
    ap.add_argument("--n", type=int, default=32, help="LR X=Y extent")
    ap.add_argument("--nz", type=int, default=128, help="vertical levels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from gan_sr_wind_field_amd import _lib, dist as wdist
    _lib.lib()  # fail loudly without the HIP extension
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    distributed = wdist.init_from_env("nccl")
    assert world == args.gpus or not distributed, f"--gpus {args.gpus} but WORLD_SIZE {world}"

    gan, cfg = make_gan(args, dev, args.dtype)
    dp = wdist.attach(gan, bucket_mb=cfg.dist.bucket_mb, sync_bn=cfg.dist.sync_bn) if distributed else None

    from oracle.gan import synthetic_batch  # input generator only (host side, before timing)
    B, n, nz, s = args.batch, args.n, args.nz, cfg.scale
    LR, HR, Z, x, y = (t.to(dev) for t in synthetic_batch(B, n, nz, s, seed=2001 + rank))
    gan.feed_xy_niter(x, y, torch.tensor(cfg.training.niter, device=dev), 1, 1)

    # ---- live timing of the dominant kernel: the N=144 implicit-GEMM (hr0 fwd + its dgrad)
    probe_events = []
    timing_on = [False]

    def probe(tag, fn):
        if timing_on[0] and tag in ("fwd:hr_convs.0.0", "dgrad:hr_convs.0.0"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            probe_events.append((e0, e1))
        else:
            fn()

    gan.G.program().launch_probe = probe

    def step(i):
        gan.optimize_parameters(LR, HR, Z, 2 * i)      # G-iteration
        gan.optimize_parameters(LR, HR, Z, 2 * i + 1)  # D-iteration
        gan.update_learning_rate()

    def barrier():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    timing_on[0] = True
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax)
    loss_ok = all(bool(torch.isfinite(v).all()) for v in gan.get_G_train_loss_dict_ref().values())
    assert loss_ok, "non-finite generator loss in the timed region"

    if rank != 0:
        return
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * 0 + world * args.steps / elapsed  # global train-steps/s (each step = `batch` samples/GPU)
    sX = s * n
    g_f, d_f = g_fwd_flops(n, nz, s), d_fwd_flops(sX, nz)
    pair_flops = B * (4 * g_f + 9 * d_f)  # G-it 3G+3D, D-it G+6D
    k_ms = [a.elapsed_time(b) for a, b in probe_events]
    k_flops = 2.0 * B * sX * sX * nz * 125 * 144 * 144
    peak = 2500.0 if args.dtype == "bf16" else 157.3
    achieved = k_flops / (sum(k_ms) / len(k_ms) * 1e-3) / 1e12 if k_ms else None
    out = {
        "metric": "GAN train-steps/sec (G+D fwd+bwd)", "value": round(value, 4), "unit": "train-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"C3' full G+D adversarial step, LR {n}x{n}x{nz} -> HR {sX}x{sX}x{nz} (x{s}), "
                               f"batch {B}/GPU, G 16 RRDB nf128 (34.77M), D 128^3 bf32",
                   "global_batch": B * world, "parallelism": f"dp{world}",
                   "step_tflop": round(pair_flops / 1e12, 2),
                   "achieved_tflops_per_gpu": round(pair_flops / 1e12 / (elapsed / args.steps), 1),
                   "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2**30, 2)},
        "roofline": {"bound": "mfma",
                     "kernel": ("conv_tile_kernel<8,1,4,9,2> (LDS halo-tile conv)" if args.dtype == "bf16"
                                else "igemm_kernel<F32,4,1,2,9>") + ": hr_convs.0 5x5x5 144->144 fwd + dgrad",
                     "achieved": round(achieved, 1) if achieved else None, "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4) if achieved else None,
                     # HBM bytes per launch from the PMC passes in profiles/r01_c_hr0_hbm_traffic_pmc.txt,
                     # corrected as MI355X_MICROARCH.md prescribes for gfx950: 2 x FETCH_SIZE (wide LDS-DMA
                     # reads are tallied at half) + WRITE_SIZE = 2 x 2.66 GB + 0.59 GB; algorithmic 1.21e9
                     "traffic": 5.92e9 if (args.dtype == "bf16" and n == 32 and nz == 128 and B == 1) else None,
                     "launches_timed": len(k_ms), "avg_launch_ms": round(sum(k_ms) / len(k_ms), 3) if k_ms else None},
    }
    if world == 1 and not args.no_cpu_baseline:
        cores = min(os.cpu_count() or 1, 16)
        dt, sample_flops = cpu_baseline(cores)
        tf_s = sample_flops / dt / 1e12
        out["cpu_baseline"] = {
            "value": round(tf_s * 1e12 / (pair_flops / B), 6), "unit": "train-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle (PyTorch CPU restatement) full-size G+D pair at 16x16x10->64x64x10 B=1: "
                      f"{dt:.2f} s/pair = {tf_s:.3f} TFLOP/s, scaled by FLOPs to the C3' step"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
