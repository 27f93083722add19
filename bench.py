#!/usr/bin/env python
"""GAN train-steps/sec of the MI355X-native wind-field GAN (BASELINE.json metric).

A step = one G-iteration + one D-iteration (reference wind_field_GAN_3D.py:585-593
alternates the two kinds) on one synthetic batch already resident in HBM.
Default workload (N = 1): BASELINE.json configs[2] in reference semantics (SURVEY 8d "C3'"):
LR (B,4,32,32,128) -> HR (B,3,128,128,128), full-size G (16 RRDB, nf 128) and the
128^3 D, bf16 compute with fp32 loss, dropout / instance noise / Adam all on.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3p|C1|C1b|C1c|C2|C3lit|C4|C5b|C5c|C5lit|C6]
                    [--dtype bf16|fp32] [--batch B] [--n N --nz NZ]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

``--gpus N`` with N > 1 and no torchrun environment starts the N ranks itself: the parent - before it makes any
GPU call - runs ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`` on this
same file as a CHILD process and exits with its code (one process per GPU over RCCL; N = 1 stays in-process so
that ``rocprofv3 -- python bench.py`` profiles the benchmark itself).  ``--dry-launch`` stops every rank after the
rendezvous (gloo, no GPU): the CPU test of the launcher.  ``--backend gloo --one-device`` rehearses the
N-rank step on ONE card (ranks share cuda:0; RCCL refuses duplicate devices, gloo does not).

``--config`` selects the other BASELINE.json configurations; they are recorded in
DESIGN.md, the driver's line is always the default C3'.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

#: name -> (ini, LR n, nz, batch/GPU, dtype, step kind, slicing, description)
PRESETS = {
    "C3p": ("local", 32, 128, 1, "bf16", "gan", False, "C3' full G+D adversarial step"),
    "C4": ("local", 32, 128, 4, "bf16", "gan", False, "C4 per-GPU shape: C3' at batch 4/GPU (global batch 32 on 8 GPUs)"),
    "C1": ("local", 16, 10, 1, "bf16", "gan", True, "C1 shipped local ini: D with slicing, 64x64x10 HR patches"),
    "C1b": ("local", 32, 10, 1, "bf16", "gan", False, "C1b the reference's real patch size, no slicing"),
    # the reference's PRODUCTION training shape: cluster ini as shipped (config/wind_field_GAN_3D_config_cluster.ini:42,47:
    # batch_size 32, enable_slicing True -> 64x64x10 HR slices of the 128x128x10 patches)
    "C1c": ("cluster", 16, 10, 32, "bf16", "gan", True, "C1c cluster ini as shipped: batch 32, D with slicing, 64x64x10 HR slices"),
    "C2": ("local", 64, 64, 1, "fp32", "g_only", False, "C2 generator-only fwd+bwd+Adam, fp32"),
    "C3lit": ("local", 128, 128, 1, "bf16", "g_only", False,
              "C3 literal reading: generator-only fwd+bwd+Adam at LR 128^3 (4.8e9-element HR tensors; D cannot take 512x512)"),
    "C5b": ("upscale8", 16, 10, 8, "bf16", "gan", False, "C5b upscale8 ini (x8, three UpConv stages), full G+D step"),
    "C5c": ("upscale8", 16, 128, 1, "bf16", "gan", False, "C5c x8 to 128^3, full G+D step"),
    "C5lit": ("upscale8", 64, 64, 1, "bf16", "g_only", False, "C5 literal: x8 generator-only fwd+bwd+Adam, HBM stress"),
    # pretrained_models/upscale16_pix4_no_adv_no_slicing/config.ini:5 (scale = 16: four UpConv stages), its batch size
    "C6": ("upscale16", 8, 10, 8, "bf16", "gan", False, "C6 upscale16 ini (x16, four UpConv stages), full G+D step"),
}


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


def d_fwd_flops(xy, nz, bf=32, slicing=False):
    """D on (3, xy, xy, nz), reference Discriminator_3D.py:55-169: z halves in block 0 when nz > 19; the last
    block is k3 s1 + k3 s(1,1,2) with slicing, k3 s1 + k(4,4,3) s2 without"""
    f, X, Z = 0, xy, nz
    cin = 3
    for i, cout in enumerate((bf, 2 * bf, 4 * bf, 8 * bf, 8 * bf)):
        f += 2 * cin * cout * 27 * X * X * Z
        if slicing and i == 4:
            Zo = (Z + 2 - 3) // 2 + 1
            f += 2 * cout * cout * 27 * X * X * Zo
        else:
            halve = (i == 0 and nz > 19) or i == 4
            Zo = (Z + 2 - 3) // 2 + 1 if halve else Z
            X //= 2
            f += 2 * cout * cout * 48 * X * X * Zo
        Z, cin = Zo, cout
    return f + 2 * (8 * bf * X * X * Z) * 100


def make_gan(args, dev, dtype):
    from gan_sr_wind_field_amd.config.config import Config
    from gan_sr_wind_field_amd.GAN_models.wind_field_GAN_3D import wind_field_GAN_3D

    cfg = Config(os.path.join(ROOT, "gan_sr_wind_field_amd", "config", f"wind_field_GAN_3D_config_{args.ini}.ini"))
    cfg.is_train, cfg.is_test, cfg.is_use = True, False, False
    cfg.gpu_id, cfg.device = dev.index, dev
    cfg.compute_dtype = dtype
    cfg.gan_config.enable_slicing = args.slicing
    cfg.gan_config.number_of_z_layers = args.nz
    cfg.training.niter = 150000
    cfg.training.d_g_train_period = 1  # it even -> G-iteration, it odd -> D-iteration
    torch.manual_seed(cfg.env.fixed_seed)
    return wind_field_GAN_3D(cfg), cfg


def granted_cores():
    """CPU cores this process may actually use: the affinity mask, cut by the cgroup's CPU quota when there is one (a
    container on a many-core host: os.cpu_count() is the HOST's count, and that many threads on a 16-core share
    oversubscribe - the oracle then runs many times slower than on the cores it really has)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(n_threads, full=False, budget_s=30.0, cap_s=90.0):
    """The oracle (CPU restatement of the reference, kind "port") timed on this host as BASELINE.md section 4 lays out:
    full-size G + D at the reference's own CPU-runnable shapes - C1 (16x16x10 -> 64x64x10, D sliced) and C1b
    (32x32x10 -> 128x128x10), B = 1, fp32, ``n_threads`` threads - G-iterations and D-iterations timed SEPARATELY,
    medians reported, TFLOP/s by the algorithmic FLOPs of BASELINE.md section 3.

    ``full``: the procedure to the letter (5 warm-up + 20 timed iterations of each kind, both shapes; minutes - run once
    per round: ``bench.py --cpu-baseline-full``, profiles/r06_cpu_baseline_full.json).  Default: a BOUNDED sample of the
    same procedure (the default run must finish in minutes): 1 warm-up pair and 5 timed pairs at each shape (C1 up to 20
    while its share of ``budget_s`` lasts) - unless the warm-up pair shows that the 5 would not fit the shape's share of
    ``cap_s`` seconds of wall clock (a noisy or slow host): then as many timed pairs as do fit, at least one, and the
    shape's record says ``"truncated": true``.  `value` is extrapolated from the C1b rate (since round 5; rounds 1-4
    used the C1 rate - the records are not comparable across that change).
    (tools/cpu_baseline_check.py times the real reference beside the oracle in the build container.)"""
    import statistics
    from oracle import gan as ogan
    from oracle import nets as onets

    torch.set_num_threads(n_threads)
    shapes = {}
    for name, n, slicing, share in (("C1", 16, True, 0.45), ("C1b", 32, False, 0.55)):
        gs = onets.GSpec(dropout_p=0.1)
        ds = onets.DSpec(bf=32, nz=10, enable_slicing=slicing, dropout_p=0.2)
        gen = torch.Generator().manual_seed(0)
        sdG, sdD = onets.make_state(onets.g_param_shapes(gs)), onets.make_state(onets.d_param_shapes(ds))
        onets.kaiming_init_(sdG, 0.1, gen)
        onets.kaiming_init_(sdD, 0.2, gen)
        gan = ogan.OracleGAN(sdG, sdD, gs, ds, ogan.TrainSpec(d_g_train_period=1))
        LR, HR, Z, x, y = ogan.synthetic_batch(1, n, 10, 4, seed=2001)
        gan.feed_xy(x, y)
        warm, lo, hi = (5, 20, 20) if full else (1, 5, 20 if name == "C1" else 5)
        tw = time.time()
        for i in range(warm):
            gan.optimize_parameters(LR, HR, Z, 2 * i)
            gan.optimize_parameters(LR, HR, Z, 2 * i + 1)
        warm_s = time.time() - tw
        truncated = False
        if not full and warm_s / warm * lo > share * cap_s:  # the hard minimum would run past the wall-clock cap
            lo = hi = max(1, int(share * cap_s / (warm_s / warm)))
            truncated = True
        print(f"[bench] cpu_baseline {name}: {warm} warm-up pair(s) {warm_s:.1f} s" + (f"; capped at {lo} timed pair(s)" if truncated else ""),
              file=sys.stderr, flush=True)
        tg, td, t0, i = [], [], time.time(), warm
        while len(tg) < lo or (len(tg) < hi and time.time() - t0 < share * budget_s):
            a = time.perf_counter()
            gan.optimize_parameters(LR, HR, Z, 2 * i)       # G-iteration
            b = time.perf_counter()
            gan.optimize_parameters(LR, HR, Z, 2 * i + 1)   # D-iteration
            c = time.perf_counter()
            tg.append(b - a)
            td.append(c - b)
            i += 1
        g_f, d_f = g_fwd_flops(n, 10), d_fwd_flops(4 * n, 10, slicing=slicing)
        mg, md = statistics.median(tg), statistics.median(td)
        shapes[name] = {"g_it_s": round(mg, 3), "d_it_s": round(md, 3), "pair_s": round(mg + md, 3),
                        "g_it_per_s": round(1 / mg, 4), "d_it_per_s": round(1 / md, 4),
                        "steps_per_s": round(1 / (mg + md), 4), "timed_pairs": len(tg), "warmup_pairs": warm,
                        "truncated": truncated,
                        "g_it_tflops": round((3 * g_f + 3 * d_f) / mg / 1e12, 3),
                        "d_it_tflops": round((g_f + 6 * d_f) / md / 1e12, 3),
                        "pair_tflops": round((4 * g_f + 9 * d_f) / (mg + md) / 1e12, 3)}
        del gan
    return shapes


TRAFFIC_JSON = "profiles/r06_hbm_traffic.json"


def recorded_traffic_commit():
    """the commit the PMC passes behind TRAFFIC_JSON were taken at (its ``_commit`` entry)"""
    return recorded_traffic("_commit")


def recorded_traffic(key):
    """HBM bytes per launch from the PMC passes of this round (TRAFFIC_JSON, written from the rocprofv3 --pmc
    summaries by tools/tuning/pmc_step_sum.py; gfx950 FETCH_SIZE correction applied there).  Recorded, not
    measured in this run: the JSON line says so (``traffic_measured_in_run: false``)."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_JSON)) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """Start ``args.gpus`` rank processes of this file through torch.distributed.run as a CHILD of this process
    (which has made no GPU call: a process that has initialised the GPU must never be replaced or forked into
    ranks) and return the child's exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(args.master_port or free_port()),
           os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    if args.one_device and args.gpus > 2:
        # rehearsal only: N processes on one card at HIP's default of four hardware queues each oversubscribe the card's
        # queue slots - with four ranks the first generator backward never finishes (profiles/README.md, round 5)
        env.setdefault("GPU_MAX_HW_QUEUES", "2")
    return subprocess.run(cmd, env=env).returncode


def dry_launch(args):
    """rendezvous only (gloo, CPU): proves that ``--gpus N`` produced N ranks; rank 0 prints the JSON line"""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        ones = torch.ones(1)
        dist.all_reduce(ones)
        ranks = dist.get_world_size()
        assert int(ones.item()) == ranks == world
        seen = [None] * world
        dist.all_gather_object(seen, (rank, int(os.environ.get("LOCAL_RANK", "0"))))
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen = [(0, 0)]
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "ranks": [list(x) for x in seen]}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3p", choices=sorted(PRESETS))
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU")
    ap.add_argument("--n", type=int, default=None, help="LR X=Y extent")
    ap.add_argument("--nz", type=int, default=None, help="vertical levels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="BASELINE.md section 4 to the letter: 5 warm-up + 20 timed G- and D-iterations at C1 and C1b (minutes)")
    ap.add_argument("--no-fp32-side", action="store_true",
                    help="skip the fp32 (reference arithmetic) side measurement of the default line")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL over xGMI")
    ap.add_argument("--one-device", action="store_true", help="every rank on cuda:0 (rehearsal with --backend gloo)")
    ap.add_argument("--dry-launch", action="store_true", help="rendezvous of the N ranks only (CPU, gloo)")
    ap.add_argument("--master-port", type=int, default=None)
    ap.add_argument("--no-comm-timing", action="store_true", help="no HIP events around the collectives")
    ap.add_argument("--single-rank-group", action="store_true",
                    help="with --gpus 1: run the data-parallel step over a process group of ONE rank (every collective "
                         "goes through RCCL; what the DP machinery costs before any second GPU)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no torchrun environment: become the launcher BEFORE anything touches the GPU
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if args.dry_launch:
        return dry_launch(args)
    if os.environ.get("WSR_BENCH_WATCHDOG"):  # debugging aid: every rank prints its stacks after this many seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["WSR_BENCH_WATCHDOG"]), repeat=False)
    # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner to stdout when a communicator
    # is made (and other libraries may chat there too) - from here on file descriptor 1 is stderr, and the line goes
    # to the saved descriptor at the end
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ini, n0, nz0, b0, dt0, kind, slicing, desc = PRESETS[args.config]
    args.ini, args.slicing = ini, slicing
    args.n = n0 if args.n is None else args.n
    args.nz = nz0 if args.nz is None else args.nz
    args.batch = b0 if args.batch is None else args.batch
    args.dtype = dt0 if args.dtype is None else args.dtype

    from gan_sr_wind_field_amd import _lib, dist as wdist
    from gan_sr_wind_field_amd.process_data import synthetic_batch
    _lib.lib()  # fail loudly without the HIP extension
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}: the line would be mislabelled"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    if args.one_device:
        local = 0
        os.environ["LOCAL_RANK"] = "0"
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    distributed = wdist.init_from_env(args.backend, single_rank=args.single_rank_group)
    assert distributed == (world > 1 or args.single_rank_group)
    ranks_answered = 1
    over_rccl = bool(distributed and args.backend == "nccl")  # (a gloo rehearsal is not evidence about RCCL)
    if distributed:  # an actual collective, not the environment: this many ranks answered
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        ranks_answered = int(ones.item())
        assert ranks_answered == torch.distributed.get_world_size() == args.gpus

    gan, cfg = make_gan(args, dev, args.dtype)
    dp = wdist.attach(gan, bucket_mb=cfg.dist.bucket_mb, sync_bn=cfg.dist.sync_bn) if distributed else None

    B, n, nz, s = args.batch, args.n, args.nz, cfg.scale
    LR, HR, Z, x, y = (t.to(dev) for t in synthetic_batch(B, n, nz, s, seed=2001 + rank))
    gan.feed_xy_niter(x, y, torch.tensor(cfg.training.niter, device=dev), 1, 1)

    # ---- live timing (HIP events on the launch stream, inside the timed region) of
    #   the dominant MFMA-bound kernel: the N=144 halo-tile conv (hr0 forward + its input gradient)
    #   the dominant HBM-bound kernel:  the streaming 1x1x1 conv (LFF forward, 256 -> 128)
    #   the memory-bound conv3d launches: the last conv in z-folded form - forward (5,5,1) 144 -> 15 and its input
    #   gradient 15 -> 144 with the LeakyReLU / Dropout3d mask of the 5x5x5 conv below it in the epilogue
    #   the rest of SURVEY 8(d)'s memory-bound conv3d set: terrain convs (1 -> 16, 16 -> 16 at HR resolution), the
    #   feature conv (4 -> 128) and the discriminator's first conv (3 -> 32, D(real) and D(fake) in one launch)
    probe_events = {"mfma": [], "hbm": [], "hbm_res2": [], "conv_dgrad": [], "conv_fwd": [], "terrain0_fwd": [],
                    "terrain1_fwd": [], "terrain1_dgrad": [], "feature_fwd": [], "d0_fwd": []}
    timing_on = [False]
    last_conv = "hr_convs.2.zfold" if gan.G.program().zfold_active() else "hr_convs.2"
    d0_name = gan.D.features.program().layers[0].conv.name if kind == "gan" else None
    side_tags = {"fwd:terrain_convs.0.0": "terrain0_fwd", "fwd:terrain_convs.1.0": "terrain1_fwd",
                 "dgrad:terrain_convs.1.0": "terrain1_dgrad", "fwd:model.0.0": "feature_fwd"}
    if d0_name:
        side_tags["fwd:" + d0_name] = "d0_fwd"

    def probe(tag, fn):
        which = None
        if timing_on[0]:
            if tag in ("fwd:hr_convs.0.0", "dgrad:hr_convs.0.0"):
                which = "mfma"
            elif tag.startswith("fwd:") and tag.endswith(".LFF"):
                # the last block of an RRDB also reads the RRDB shortcut (a second residual: 128 more channels)
                which = "hbm_res2" if tag.endswith(".RDBs.2.LFF") else "hbm"
            elif tag == "dgrad:" + last_conv:
                which = "conv_dgrad"
            elif tag == "fwd:" + last_conv:
                which = "conv_fwd"
            else:
                which = side_tags.get(tag)
        if which is None:
            fn()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        probe_events[which].append((e0, e1))

    gan.G.program().launch_probe = probe
    if kind == "gan":
        gan.D.features.program().launch_probe = probe

    if kind == "gan":
        def step(i):
            gan.optimize_parameters(LR, HR, Z, 2 * i)      # G-iteration
            gan.optimize_parameters(LR, HR, Z, 2 * i + 1)  # D-iteration
            gan.update_learning_rate()
    else:  # generator only: forward, L1 loss against HR, backward, Adam (reference pixel criterion)
        l1 = torch.nn.L1Loss()

        def step(i):
            gan.G.train()
            gan.optimizer_G.zero_grad(set_to_none=True)
            l1(gan.G(LR, Z), HR).backward()
            gan.optimizer_G.step()

    def barrier():
        if distributed:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    timing_on[0] = True
    if dp is not None:
        dp.stats.reset()
        dp.stats.timing = not args.no_comm_timing
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    issued = time.perf_counter() - t0  # host time to issue the steps (the loss guards sync once per G-iteration)
    barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = [elapsed / args.steps * 1e3]
    if distributed:
        mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(torch.distributed.get_world_size())]
        torch.distributed.all_gather(every, mine)
        rank_ms = [float(t) / args.steps * 1e3 for t in every]  # each rank's own clock around the same K steps
        elapsed = max(float(t) for t in every)  # the step time the line reports: MAX over ranks
    if kind == "gan":
        loss_ok = all(bool(torch.isfinite(v).all()) for v in gan.get_G_train_loss_dict_ref().values())
        assert loss_ok, "non-finite generator loss in the timed region"

    def leave():  # together: a rank that exits while another still waits in a collective hangs the launcher
        if distributed:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()

    if rank != 0:
        leave()
        return
    ms_per_step = elapsed / args.steps * 1e3
    # One data-parallel step is ONE global optimiser step over world * B samples (weak scaling: B per GPU is
    # fixed).  The metric counts train-steps at the quoted per-GPU shape, so the whole-job aggregate is the
    # sample-normalised rate: world * (B / B_quoted) / step time, with B_quoted = this preset's batch.
    steps_per_s = args.steps / elapsed
    value = world * steps_per_s
    sX = s * n
    g_f = g_fwd_flops(n, nz, s)
    d_f = d_fwd_flops(sX, nz, slicing=slicing) if kind == "gan" else 0
    step_flops = B * ((4 * g_f + 9 * d_f) if kind == "gan" else 3 * g_f)  # G-it 3G+3D, D-it G+6D
    peak = 2500.0 if args.dtype == "bf16" else 157.3
    esz = 2 if args.dtype == "bf16" else 4

    def mean_ms(evs):
        ms = [a.elapsed_time(b) for a, b in evs]
        return (sum(ms) / len(ms), len(ms)) if ms else (None, 0)

    k_ms, k_n = mean_ms(probe_events["mfma"])
    k_flops = 2.0 * B * sX * sX * nz * 125 * 144 * 144
    achieved = k_flops / (k_ms * 1e-3) / 1e12 if k_ms else None
    # LFF forward, algorithmic bytes: the 256-channel dense buffer read once + 128 channels written +
    # the filter.  The block's identity shortcut is the first 128 channels of that same input - the lane's own
    # K-fragment, no second read.  Every third launch (the last block of an RRDB) also reads the RRDB
    # shortcut: 128 more channels.  achieved = all algorithmic bytes of the timed launches / their total time.
    vox = B * n * n * nz
    h_b1 = vox * (256 + 128) * esz + 256 * 128 * esz
    h_b2 = h_b1 + vox * 128 * esz
    h1 = [a.elapsed_time(b) for a, b in probe_events["hbm"]]
    h2 = [a.elapsed_time(b) for a, b in probe_events["hbm_res2"]]
    h_n = len(h1) + len(h2)
    h_ms = (sum(h1) + sum(h2)) / h_n if h_n else None
    h_bytes = (len(h1) * h_b1 + len(h2) * h_b2) / h_n if h_n else None
    h_ach = h_bytes / (h_ms * 1e-3) / 1e9 if h_ms else None
    # the memory-bound conv3d pair (reference Generator_3D_Resnet_ESRGAN.py:105-110, hr_convs[2]): per launch the
    # 144-channel HR tensor (read by the forward; written by the input gradient, which also reads the saved
    # 144-channel output of hr_convs[0] for its LeakyReLU / Dropout3d mask) + the 3-channel side as stored
    # (z-folded: 15 planar fp32 partial sums forward, 16 bf16 channels backward) + the filter
    V = B * sX * sX * nz
    kz_fold = gan.G.program().hr1.kernel[2] if gan.G.program().zfold_active() else 1
    c_hr = 144
    cf_bytes = V * (c_hr * esz + 3 * kz_fold * 4) + 125 * c_hr * 3 * esz
    cd_bytes = V * ((3 * kz_fold + 7) // 8 * 8 * esz + 2 * c_hr * esz) + 125 * c_hr * 3 * esz

    def hbm_block(evs, nbytes, kernel, flops=0.0):
        """HBM roofline of one memory-bound conv launch; ``flops`` (algorithmic, 2 x MACs) adds the OTHER roofline
        beside it - the time the launch's arithmetic takes at the dtype's dense MFMA peak - and says which of the
        two floors is the higher one: in fp32 (matrix rate 1/16 of bf16) most of this set is matrix-bound - the last
        conv's 0.48 TFLOP at C2 take 3.1 ms at the fp32 MFMA peak against 0.33 ms at 8 TB/s"""
        ms_, n_ = mean_ms(evs)
        ach = nbytes / (ms_ * 1e-3) / 1e9 if ms_ else None
        blk = {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 1) if ach else None, "peak": 8000.0,
               "unit": "GB/s", "frac": round(ach / 8000.0, 4) if ach else None, "traffic": None,
               "algorithmic_bytes": int(nbytes), "launches_timed": n_,
               "avg_launch_us": round(ms_ * 1e3, 2) if ms_ else None}
        if flops and ms_:
            hbm_us, mfma_us = nbytes / 8e12 * 1e6, flops / (peak * 1e12) * 1e6
            blk.update({"algorithmic_gflop": round(flops / 1e9, 2), "hbm_floor_us": round(hbm_us, 2),
                        "mfma_floor_us": round(mfma_us, 2), "binding_roofline": "mfma" if mfma_us > hbm_us else "hbm",
                        "frac_of_binding_roofline": round(max(hbm_us, mfma_us) / (ms_ * 1e3), 4)})
            if achieved:
                # ... and the same two floors with the matrix side priced at a rate this chip has been SEEN to hold - the
                # 5x5x5 conv's own measured rate in this run - instead of the nominal peak: a launch whose arithmetic takes
                # longer than its bytes at that rate is co-bound, and 0.60 of the HBM figure is then out of its reach
                ach_us = flops / (achieved * 1e12) * 1e6
                blk.update({"mfma_floor_us_at_achieved_rate": round(ach_us, 2), "achieved_mfma_rate_tflops": round(achieved, 1),
                            "binding_roofline_at_achieved_rate": "mfma" if ach_us > hbm_us else "hbm",
                            "frac_of_binding_roofline_at_achieved_rate": round(max(hbm_us, ach_us) / (ms_ * 1e3), 4),
                            "hbm_frac_ceiling_at_achieved_rate": round(min(1.0, hbm_us / ach_us), 4)})
        return blk

    hr1_flops = 2.0 * V * 125 * c_hr * 3  # (the fold computes 16 columns for 15: counted as the conv's 3 x 125 taps)
    conv_d = hbm_block(probe_events["conv_dgrad"], cd_bytes,
                       "hr_convs.2 input gradient (z-folded 15 -> 144, 5x5x1, LeakyReLU+Dropout3d mask epilogue)", hr1_flops)
    conv_f = hbm_block(probe_events["conv_fwd"], cf_bytes, "hr_convs.2 forward (z-folded 144 -> 15, 5x5x1, planar fp32 out)",
                       hr1_flops)
    default_shape = args.dtype == "bf16" and n == 32 and nz == 128 and B == 1 and s == 4
    if default_shape:
        conv_d["traffic"], conv_f["traffic"] = recorded_traffic("hr1_dgrad"), recorded_traffic("hr1_fwd")
    # the other members of the memory-bound conv3d set: (Cin * V_in + Cout * V_out) * sizeof + filter
    tf_c, nf_c = gan.G.program().tf, gan.G.program().nf
    side = [
        ("terrain0_fwd", "terrain_convs.0 forward (3x3x3, 1 -> %d, LeakyReLU)" % tf_c, V * (1 + tf_c) * esz + 27 * tf_c * esz,
         2.0 * V * 27 * tf_c),
        ("terrain1_fwd", "terrain_convs.1 forward (3x3x3, %d -> %d, the terrain half of the concat)" % (tf_c, tf_c),
         V * 2 * tf_c * esz + 27 * tf_c * tf_c * esz, 2.0 * V * 27 * tf_c * tf_c),
        ("terrain1_dgrad", "terrain_convs.1 input gradient (3x3x3, %d -> %d)" % (tf_c, tf_c),
         V * 2 * tf_c * esz + 27 * tf_c * tf_c * esz, 2.0 * V * 27 * tf_c * tf_c),
        ("feature_fwd", "model.0 feature conv forward (3x3x3, 4 -> %d)" % nf_c, vox * (4 + nf_c) * esz + 27 * 4 * nf_c * esz,
         2.0 * vox * 27 * 4 * nf_c),
    ]
    if kind == "gan":
        d0 = gan.D.features.program().layers[0].conv
        side.append(("d0_fwd", "D features.0.0 forward (3x3x3, %d -> %d, D(real) + D(fake) in one launch)" % (d0.cin, d0.cout),
                     2 * V * (d0.cin + d0.cout) * esz + 27 * d0.cin * d0.cout * esz, 2.0 * 2 * V * 27 * d0.cin * d0.cout))
    side_blocks = []
    for key, name, nbytes, flops in side:
        blk = hbm_block(probe_events[key], nbytes, name, flops)
        if default_shape:
            blk["traffic"] = recorded_traffic(key)
        side_blocks.append(blk)
    out = {
        "metric": "GAN train-steps/sec (G+D fwd+bwd)" if kind == "gan" else "generator train-steps/sec (G fwd+bwd+Adam)",
        "value": round(value, 4), "unit": "train-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        # `value` is the whole-job aggregate the contract asks for under weak scaling: n_gpus x (steps/s at the quoted
        # per-GPU shape) = samples/s / B.  The north_star's ">= 10 train-steps/s at batch 8 x 128^3 on 8 GPUs" counts
        # GLOBAL optimiser steps (one step = all ranks' samples): that quantity is `global_steps_per_s`.
        "global_steps_per_s": round(steps_per_s, 4),
        "value_definition": "n_gpus x global_steps_per_s (sample-normalised whole-job rate, weak scaling)",
        # (judged only where it is defined: 8 ranks over RCCL at the C3' per-GPU shape; null anywhere else)
        "north_star_target": {"quantity": "global_steps_per_s", "target": 10.0, "at": "n_gpus = 8, batch 1/GPU (C3')",
                              "met": bool(steps_per_s >= 10.0) if (args.config == "C3p" and world == 8 and over_rccl) else None},
        "ranks": ranks_answered, "rccl_ranks": ranks_answered if over_rccl else None,
        "backend": args.backend if distributed else None,
        "rank_ms_per_step": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3),
                             "mean": round(sum(rank_ms) / len(rank_ms), 3), "per_rank": [round(v, 3) for v in rank_ms]},
        "config": {"workload": f"{desc}: LR {n}x{n}x{nz} -> HR {sX}x{sX}x{nz} (x{s}), batch {B}/GPU, "
                               f"G 16 RRDB nf128 (34.77M)" + (f", D bf32{' sliced' if slicing else ''}" if kind == "gan" else ""),
                   "preset": args.config, "global_batch": B * world, "parallelism": f"dp{world}",
                   "ranks": ranks_answered, "rccl_ranks": ranks_answered if over_rccl else None,
                   "backend": args.backend if distributed else None,
                   "global_steps_per_s": round(steps_per_s, 4), "samples_per_s": round(world * B * steps_per_s, 4),
                   "step_tflop": round(step_flops / 1e12, 2),
                   "achieved_tflops_per_gpu": round(step_flops / 1e12 / (elapsed / args.steps), 1),
                   "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2**30, 2),
                   "host_issue_ms_per_step": round(issued / args.steps * 1e3, 2)},
        "roofline": {"bound": "mfma",
                     "kernel": ("conv_tile_kernel<8,1,4,9,2> (LDS halo-tile conv)" if args.dtype == "bf16"
                                else "igemm_kernel<F32,4,1,2,9>") + ": hr_convs.0 5x5x5 144->144 fwd + dgrad",
                     "achieved": round(achieved, 1) if achieved else None, "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4) if achieved else None,
                     "traffic": recorded_traffic("hr0") if default_shape else None,
                     "traffic_source": TRAFFIC_JSON + " (PMC passes of this round)" if default_shape else None,
                     "traffic_measured_in_run": False,
                     "traffic_measured_at_commit": recorded_traffic_commit() if default_shape else None,
                     "launches_timed": k_n, "avg_launch_ms": round(k_ms, 3) if k_ms else None},
        "roofline_hbm": {"bound": "hbm", "kernel": "conv1x1_kernel (streaming 1x1x1 GEMM): RDB LFF 256->128 fwd + residuals",
                         "achieved": round(h_ach, 1) if h_ach else None, "peak": 8000.0, "unit": "GB/s",
                         "frac": round(h_ach / 8000.0, 4) if h_ach else None,
                         "traffic": recorded_traffic("lff_fwd") if default_shape else None,
                         "traffic_measured_in_run": False,
                         "algorithmic_bytes": int(h_bytes) if h_bytes else None, "launches_timed": h_n,
                         "launches_with_second_residual": len(h2),
                         "avg_launch_us": round(h_ms * 1e3, 2) if h_ms else None},
        # the WORST of the memory-bound conv3d launches carries the name; all of them are listed
        "roofline_hbm_conv": dict(min([conv_d, conv_f] + side_blocks,
                                      key=lambda r: r["frac"] if r["frac"] is not None else 9.0),
                                  traffic_measured_in_run=False),
        "roofline_hbm_convs": [conv_d, conv_f] + side_blocks,
    }
    if dp is not None:
        out["comm"] = dp.stats.summary(args.steps)
        out["comm"]["bucket_mb"] = cfg.dist.bucket_mb
        out["comm"]["sync_bn"] = bool(cfg.dist.sync_bn)
    if world == 1 and not args.no_fp32_side and args.dtype == "bf16" and kind == "gan" and args.config == "C3p":
        # The reference computes in fp32 (its AMP lines are commented out, Generator_3D_Resnet_ESRGAN.py:65): the same
        # workload in that arithmetic, a few steps, reported beside the bf16 line (never as `value`).
        del step
        gan.G.program().launch_probe = None
        gan.D.features.program().launch_probe = None
        del gan
        torch.cuda.empty_cache()
        gan32, _ = make_gan(args, dev, "fp32")
        gan32.feed_xy_niter(x, y, torch.tensor(cfg.training.niter, device=dev), 1, 1)

        def step32(i):
            gan32.optimize_parameters(LR, HR, Z, 2 * i)
            gan32.optimize_parameters(LR, HR, Z, 2 * i + 1)
            gan32.update_learning_rate()

        n32 = 3
        for i in range(2):
            step32(i)
        torch.cuda.synchronize()
        t32 = time.perf_counter()
        for i in range(2, 2 + n32):
            step32(i)
        torch.cuda.synchronize()
        ms32 = (time.perf_counter() - t32) / n32 * 1e3
        tf32 = step_flops / 1e12 / (ms32 * 1e-3)
        out["fp32_reference_arithmetic"] = {
            "ms_per_step": round(ms32, 2), "steps": n32, "warmup": 2, "train_steps_per_s": round(1e3 / ms32, 4),
            "achieved_tflops": round(tf32, 1), "peak": 157.3, "frac": round(tf32 / 157.3, 4),
            "note": "same workload with compute_dtype fp32 (v_mfma_f32_16x16x4_f32 tile kernels): the reference's own arithmetic"}
        del gan32
        torch.cuda.empty_cache()
    if world == 1 and not args.no_cpu_baseline:
        # all cores this process may run on ("all physical cores, count stated")
        cores = granted_cores()
        print(f"[bench] cpu_baseline: oracle on {cores} threads (host reports {os.cpu_count()} CPUs) ...", file=sys.stderr,
              flush=True)
        shapes = cpu_baseline(cores, full=args.cpu_baseline_full)
        # this workload's step on the CPU, extrapolated by FLOPs (last bullet) from the LARGER
        # measured shape (C1b: the rate a CPU holds on the bigger volume; C1 is printed beside it)
        tf_s = shapes["C1b"]["pair_tflops"]
        try:
            with open("/proc/cpuinfo") as f:
                cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), None)
        except OSError:
            cpu_model = None
        out["cpu_baseline"] = {
            "value": round(tf_s * 1e12 / (step_flops / B), 6), "unit": "train-steps/s", "cores": cores, "host_cpus": os.cpu_count(),
            "cpu_model": cpu_model, "kind": "port", "procedure": "BASELINE.md section 4" + (
                "" if args.cpu_baseline_full else ", bounded sample (full: --cpu-baseline-full)"),
            "shapes": shapes,
            "sample": f"oracle (PyTorch CPU restatement, fp32) full-size G+D at B=1, G- and D-iterations timed separately, "
                      f"medians: C1 16x16x10->64x64x10 {shapes['C1']['pair_s']:.2f} s/pair = {shapes['C1']['steps_per_s']} steps/s "
                      f"({shapes['C1']['pair_tflops']} TFLOP/s, {shapes['C1']['timed_pairs']} pairs); C1b 32x32x10->128x128x10 "
                      f"{shapes['C1b']['pair_s']:.2f} s/pair = {shapes['C1b']['steps_per_s']} steps/s ({tf_s} TFLOP/s, "
                      f"{shapes['C1b']['timed_pairs']} pairs); `value` = this workload's step extrapolated by FLOPs at the C1b rate"}
    print(json.dumps(out), file=json_out, flush=True)
    leave()


if __name__ == "__main__":
    main()
