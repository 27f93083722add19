"""Data-parallel path on the CPU: world_size = 2 over gloo (``dist.py``).

The collectives that make an N-rank step equal the single-process step on the
concatenated batch - RaGAN batch means (with their backward), the max-reduced
physics-loss normalisers, bucketed gradient averaging - are exercised with the
product ``wind_field_GAN_3D`` whose networks are answered by the CPU oracle
(``tests/oracle_nets.py``); the HIP programs plug into the same hooks on the GPU.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO

LOCAL_INI = os.path.join(REPO, "gan_sr_wind_field_amd", "config", "wind_field_GAN_3D_config_local.ini")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build_gan():
    import oracle_nets
    from gan_sr_wind_field_amd.config.config import Config
    from gan_sr_wind_field_amd.GAN_models import wind_field_GAN_3D as mod
    from oracle import nets as onets

    mod.Generator_3D = oracle_nets.OracleGenerator
    mod.Discriminator_3D = oracle_nets.OracleDiscriminator
    cfg = Config(LOCAL_INI)
    cfg.is_train, cfg.is_test, cfg.is_use = True, False, False
    cfg.gpu_id, cfg.device = None, torch.device("cpu")
    cfg.generator.num_features, cfg.generator.num_RRDB, cfg.generator.RDB_growth_chan = 16, 1, 8
    cfg.generator.terrain_number_of_features = 4
    cfg.generator.dropout_probability = cfg.discriminator.dropout_probability = 0.0
    cfg.discriminator.num_features = 4
    cfg.gan_config.number_of_z_layers = 4
    cfg.training.use_instance_noise = False
    cfg.training.use_noisy_labels = False
    cfg.training.niter = 150000
    torch.manual_seed(2001)
    gan = mod.wind_field_GAN_3D(cfg)
    gs = onets.GSpec(in_channels=4, nf=16, n_rrdb=1, gc=8, tf=4, hr_kern=5, upscale=4)
    ds = onets.DSpec(bf=4, nz=4, enable_slicing=True)
    gan.G.load_state_dict(onets.deterministic_state(onets.g_param_shapes(gs), seed=41, scale=0.5))
    gan.D.load_state_dict(onets.deterministic_state(onets.d_param_shapes(ds), seed=43, scale=1.0))
    return gan, cfg


def _g_iteration(gan, cfg, LR, HR, Z, x, y):
    gan.feed_xy_niter(x, y, torch.tensor(cfg.training.niter), 1, 1)
    gan.optimize_parameters(LR, HR, Z, 0)  # it = 0 -> G-iteration (D in eval mode: no batch statistics)
    sd = gan.G.state_dict()
    keys = ("model.0.0.weight", "hr_convs.2.weight", "hr_convs.0.0.weight", "model.1.module.0.RDBs.1.LFF.bias")
    losses = {k: float(v) for k, v in gan.get_G_train_loss_dict_ref().items()}
    return {k: sd[k].clone() for k in keys}, losses


def _worker(rank, world, port, out_dir, hr_scale=1.0):
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import torch.distributed as dist
    from gan_sr_wind_field_amd import dist as wdist
    from oracle.gan import synthetic_batch

    assert wdist.init_from_env("gloo")
    gan, cfg = _build_gan()
    dp = wdist.attach(gan, bucket_mb=0.05, sync_bn=True)
    LR, HR, Z, x, y = synthetic_batch(world, 16, 4, 4, seed=2001)
    HR = HR * hr_scale
    sl = slice(rank, rank + 1)  # one sample per rank
    w, losses = _g_iteration(gan, cfg, LR[sl], HR[sl], Z[sl], x, y)
    # primitives
    t = torch.tensor([1.0 + rank, 3.0 * (rank + 1)], requires_grad=True)
    m = dp.batch_mean(t)
    (m * m).backward()
    mx = dp.global_max(torch.tensor([float(rank), -float(rank)]))
    torch.save({"w": w, "losses": losses, "mean": float(m), "mean_grad": t.grad.clone(), "max": mx,
                "n_coll": dp.n_collectives}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_g_iteration_sr_normaliser_branch(tmp_path):
    """HR scaled by 1e-4: every physics normaliser is SR_max / 100, the branch in which the reference
    differentiates the maxima (wind_field_GAN_3D.py:773-814).  Under data parallelism the maxima are max-reduced
    and the gradient goes to the rank and element that own them (dist._GlobalMax): the 2-rank G-iteration still
    equals the single-process step on the full batch."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from oracle.gan import synthetic_batch

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), 1e-4), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    gan, cfg = _build_gan()
    LR, HR, Z, x, y = synthetic_batch(world, 16, 4, 4, seed=2001)
    HR = HR * 1e-4
    # the normalisers really come from SR here
    from gan_sr_wind_field_amd.process_data import calculate_gradient_of_wind_field as jac
    with torch.no_grad():
        sr = gan.G(LR, Z)
        g_hr, g_sr = jac(HR, x, y, Z), jac(sr, x, y, Z)
    assert float(g_sr[:, :6].abs().max()) / 100 > float(g_hr[:, :6].abs().max())
    w_ref, losses_ref = _g_iteration(gan, cfg, LR, HR, Z, x, y)
    for k, v in w_ref.items():
        assert torch.equal(r0["w"][k], r1["w"][k]), k
        np.testing.assert_allclose(r0["w"][k].numpy(), v.numpy(), rtol=2e-5, atol=1e-7, err_msg=k)
    # the normalised terms are batch-global, so both ranks log the full-batch value
    np.testing.assert_allclose(0.5 * (r0["losses"]["xy_gradient"] + r1["losses"]["xy_gradient"]),
                               losses_ref["xy_gradient"], rtol=1e-4)


def test_two_rank_g_iteration_equals_full_batch(tmp_path):
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from oracle.gan import synthetic_batch

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")

    gan, cfg = _build_gan()
    LR, HR, Z, x, y = synthetic_batch(world, 16, 4, 4, seed=2001)
    w_ref, losses_ref = _g_iteration(gan, cfg, LR, HR, Z, x, y)
    for k, v in w_ref.items():
        # both ranks end with the same weights, equal to the single-process step on the full batch
        assert torch.equal(r0["w"][k], r1["w"][k]), k
        np.testing.assert_allclose(r0["w"][k].numpy(), v.numpy(), rtol=2e-5, atol=1e-7, err_msg=k)
    # batch-mean losses: the average over ranks of the per-rank means is the full-batch mean for the
    # terms that are plain means over samples (pix); adversarial uses the global logit means
    np.testing.assert_allclose(0.5 * (r0["losses"]["pix"] + r1["losses"]["pix"]), losses_ref["pix"], rtol=1e-5)
    np.testing.assert_allclose(0.5 * (r0["losses"]["adversarial"] + r1["losses"]["adversarial"]),
                               losses_ref["adversarial"], rtol=1e-4)
    # primitives: mean over the 4 values {1, 3, 2, 6} = 3; d(m^2)/dt_i = 2 m / 4 per rank, summed over ranks' losses
    assert abs(r0["mean"] - 3.0) < 1e-6 and abs(r1["mean"] - 3.0) < 1e-6
    np.testing.assert_allclose(r0["mean_grad"].numpy(), np.full(2, 2 * 2 * 3.0 / 4), rtol=1e-6)
    assert r0["max"].tolist() == [1.0, 0.0]
    assert r0["n_coll"] >= 2  # more than one gradient bucket was all-reduced


def _run_bench(argv, env_extra=None, timeout=180):
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_launcher_starts_one_rank_per_gpu():
    """``bench.py --gpus N`` without a torchrun environment must itself become N ranks (children of a parent that
    never touched the GPU) - a driver that runs ``python bench.py --gpus 2`` must not get a dp1 line."""
    import json
    r = _run_bench(["--gpus", "2", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["dry_launch"] and out["n_gpus"] == 2
    assert sorted(map(tuple, out["ranks"])) == [(0, 0), (1, 1)]  # (RANK, LOCAL_RANK) of the two children
    # N = 1 stays in-process (rocprofv3 profiles the benchmark itself, not a launcher)
    r1 = _run_bench(["--dry-launch"])
    assert r1.returncode == 0 and json.loads(r1.stdout.splitlines()[-1])["ranks"] == [[0, 0]]


def test_bench_launcher_rendezvous_of_eight_ranks():
    """the driver's N = 8 command line, as far as it can be rehearsed without GPUs: ``--gpus 8`` becomes eight ranks
    that all answer one all-reduce (gloo) with distinct (RANK, LOCAL_RANK) pairs, and rank 0 prints the one line"""
    import json
    r = _run_bench(["--gpus", "8", "--dry-launch"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    out = json.loads(lines[0])
    assert out["dry_launch"] and out["n_gpus"] == 8
    assert sorted(map(tuple, out["ranks"])) == [(i, i) for i in range(8)]


def test_bench_refuses_a_mislabelled_world():
    """inside a torchrun environment whose WORLD_SIZE differs from --gpus the run stops instead of printing a
    line with the wrong n_gpus"""
    r = _run_bench(["--gpus", "2", "--dry-launch"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def _ledger_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from gan_sr_wind_field_amd import dist as wdist

    assert wdist.init_from_env("gloo")
    dp = wdist.DataParallel(bucket_mb=4 * 100 / 2**20)  # buckets of 100 floats
    flat = torch.full((450,), float(rank + 1))
    for lo in range(0, 450, 50):
        dp.grad_ready("G", flat, lo, lo + 50)
    dp.grad_done("G")
    stat = dp.stat_allgather(torch.tensor([float(rank), 1.0]))
    s = torch.ones(6)
    dp.stat_allreduce(s)
    m = dp.global_max(torch.tensor([float(rank)]))
    mean = dp.batch_mean(torch.tensor([float(rank), float(rank)]))
    # both RaGAN average logits in ONE collective per pass (forward and backward)
    a = torch.tensor([float(rank), 2.0 * rank], requires_grad=True)
    b = torch.tensor([1.0 + rank], requires_grad=True)
    ma, mb = dp.batch_means(a, b)
    (3.0 * ma + 5.0 * mb).backward()
    comm = dp.stats.summary(1)
    # buckets shrink towards the end of the pass (tail_mb): a second ledger on its own would be the same object here,
    # so the sizes are read off the collectives' tensors
    dp2 = wdist.DataParallel(bucket_mb=4 * 100 / 2**20, tail_mb=4 * 10 / 2**20)
    sizes = []
    avg = dp2._avg_async
    dp2._avg_async = lambda t: (sizes.append(t.numel()), avg(t))[1]
    flat2 = torch.full((450,), float(rank + 1))
    for lo in range(0, 450, 10):
        dp2.grad_ready("G", flat2, lo, lo + 10)
    dp2.grad_done("G")
    torch.save(dict(flat=flat, stat=stat, s=s, m=m, mean=mean, comm=comm, ma=ma.detach(), mb=mb.detach(), ga=a.grad,
                    gb=b.grad, sizes=sizes, flat2=flat2), os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_comm_ledger_counts_collectives_and_bytes(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_ledger_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = torch.load(tmp_path / "r0.pt")
    assert torch.allclose(r["flat"], torch.full((450,), 1.5))  # averaged over the ranks
    assert r["stat"].shape == (2, 2) and r["stat"][:, 0].tolist() == [0.0, 1.0]
    assert r["s"].tolist() == [2.0] * 6 and float(r["m"]) == 1.0 and float(r["mean"]) == 0.5
    c = r["comm"]
    assert c["grad_bucket_collectives_per_step"] == 5            # 4 full buckets of 100 + the 50-float tail
    assert abs(c["grad_mbytes_per_step"] - 450 * 4 / 1e6) < 1e-3
    assert c["syncbn_collectives_per_step"] == 2
    assert c["scalar_collectives_per_step"] == 4                 # global_max, batch_mean, batch_means forward + backward
    assert c["collectives_per_step"] == 11 and c["timed"] is False and c["exposed_grad_wait_ms_per_step"] is None
    # means over the global batch (rank 0: a = [0, 0], b = [1]; rank 1: a = [1, 2], b = [2]) and their gradients:
    # d(3 ma + 5 mb) summed over both ranks' losses = 6 / 4 per element of a, 10 / 2 per element of b
    assert abs(float(r["ma"]) - 0.75) < 1e-6 and abs(float(r["mb"]) - 1.5) < 1e-6
    assert torch.allclose(r["ga"], torch.full((2,), 1.5)) and torch.allclose(r["gb"], torch.full((1,), 5.0))
    # geometric tail: full buckets, then each bucket at least as large as what is still to come; the last one is tiny
    assert r["sizes"] == [100, 100, 100, 80, 40, 20, 10] and torch.allclose(r["flat2"], torch.full((450,), 1.5))


def _means_max_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from gan_sr_wind_field_amd import dist as wdist

    assert wdist.init_from_env("gloo")
    dp = wdist.DataParallel()
    a = torch.tensor([float(rank), 2.0 * rank], requires_grad=True)
    b = torch.tensor([1.0 + rank], requires_grad=True)
    m = torch.tensor([[1.0 + rank, 5.0 - rank, 0.5, 2.0 * rank], [3.0, float(rank), 7.0 - 3 * rank, 1.0]])
    if rank == 1:
        m[1, 3] = float("nan")  # a NaN maximum on one rank reaches every rank (torch.max semantics)
    ma, mb, gm = dp.means_and_max(a, b, m)
    assert not gm.requires_grad
    got = []
    dp.ride(torch.tensor([0.0, float(rank), 0.0]), lambda v: got.append(v.clone()))  # flag 1 set on rank 1 only
    (3.0 * ma + 5.0 * mb).backward()
    assert dp.take_unridden() is None  # the backward collective of the means carried it
    # a backward pass without any scalar collective leaves the rider to its owner
    dp.ride(torch.tensor([1.0]), lambda v: None)
    left = dp.take_unridden()
    comm = dp.stats.summary(1)
    torch.save(dict(ma=ma.detach(), mb=mb.detach(), gm=gm, ga=a.grad, gb=b.grad, flags=got[0], left=left[0], comm=comm),
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_means_and_maxima_share_one_collective_and_flags_ride_on_the_backward_one(tmp_path):
    """dist._MeansMax: the generator iteration's RaGAN mean logits and physics-loss maxima in ONE all-gather; its
    backward all-reduce also carries the guard flags (``DataParallel.ride``): 2 scalar collectives where there were 4."""
    world, port = 2, _free_port()
    mp.spawn(_means_max_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        r = torch.load(tmp_path / f"r{rank}.pt")
        assert abs(float(r["ma"]) - 0.75) < 1e-6 and abs(float(r["mb"]) - 1.5) < 1e-6
        assert torch.allclose(r["ga"], torch.full((2,), 1.5)) and torch.allclose(r["gb"], torch.full((1,), 5.0))
        want = torch.tensor([[2.0, 5.0, 0.5, 2.0], [3.0, 1.0, 7.0, float("nan")]])
        assert torch.allclose(r["gm"], want, equal_nan=True), r["gm"]
        assert [bool(v > 0) for v in r["flags"]] == [False, True, False]
        assert r["left"].tolist() == [1.0]
        assert r["comm"]["scalar_collectives_per_step"] == 2 and r["comm"]["collectives_per_step"] == 2
