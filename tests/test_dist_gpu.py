"""Data-parallel train step through the HIP programs: two ranks (both on cuda:0, gloo transport so
that one card is enough) against the single-process step on the concatenated batch.

Covers what the CPU gloo test cannot: gradient buckets filled by the programs' hand-written backward,
SyncBN statistics (forward and backward sums) inside the discriminator program, and the D-iteration.
On a multi-GPU node the same hooks run over backend "nccl" (RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu
LOCAL_INI = os.path.join(REPO, "gan_sr_wind_field_amd", "config", "wind_field_GAN_3D_config_local.ini")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build_gan(dtype, device_index=0):
    from gan_sr_wind_field_amd.config.config import Config
    from gan_sr_wind_field_amd.GAN_models.wind_field_GAN_3D import wind_field_GAN_3D
    from oracle import nets as onets

    dev = torch.device(f"cuda:{device_index}")
    cfg = Config(LOCAL_INI)
    cfg.is_train, cfg.is_test, cfg.is_use = True, False, False
    cfg.gpu_id, cfg.device = device_index, dev
    cfg.compute_dtype = dtype
    cfg.generator.num_features, cfg.generator.num_RRDB, cfg.generator.RDB_growth_chan = 16, 1, 8
    cfg.generator.terrain_number_of_features = 8
    cfg.generator.dropout_probability = cfg.discriminator.dropout_probability = 0.0
    cfg.discriminator.num_features = 8
    cfg.gan_config.number_of_z_layers = 4
    cfg.training.use_instance_noise = False
    cfg.training.use_noisy_labels = False
    cfg.training.niter = 150000
    torch.manual_seed(2001)
    gan = wind_field_GAN_3D(cfg)
    gs = onets.GSpec(in_channels=4, nf=16, n_rrdb=1, gc=8, tf=8, hr_kern=5, upscale=4)
    ds = onets.DSpec(bf=8, nz=4, enable_slicing=True)
    gan.G.load_state_dict(onets.deterministic_state(onets.g_param_shapes(gs), seed=41, scale=0.5))
    gan.D.load_state_dict(onets.deterministic_state(onets.d_param_shapes(ds), seed=43, scale=1.0))
    return gan, cfg


def _two_iterations(gan, cfg, LR, HR, Z, x, y):
    dev = cfg.device
    gan.feed_xy_niter(x.to(dev), y.to(dev), torch.tensor(cfg.training.niter, device=dev), 1, 1)
    keys_g = ("model.0.0.weight", "hr_convs.2.weight", "hr_convs.0.0.weight", "model.1.module.0.RDBs.1.LFF.bias")
    keys_d = ("features.0.0.0.weight", "features.1.1.1.weight", "features.3.1.0.weight", "classifier.2.weight",
              "features.2.0.1.running_var")
    gan.optimize_parameters(LR.to(dev), HR.to(dev), Z.to(dev), 0)  # G-iteration
    # the (averaged) gradients that drove the Adam step - what the buckets carried, before any sign(g) step hides a
    # mis-scaled or dropped bucket behind "the weights moved by about lr"
    out = {"gradG." + k: dict(gan.G.named_parameters())[k].grad.detach().float().cpu().clone() for k in keys_g}
    gan.optimize_parameters(LR.to(dev), HR.to(dev), Z.to(dev), 1)  # D-iteration (train-mode BatchNorm)
    out.update({"gradD." + k: dict(gan.D.named_parameters())[k].grad.detach().float().cpu().clone()
                for k in keys_d if "running" not in k})
    sdG, sdD = gan.G.state_dict(), gan.D.state_dict()
    out.update({"G." + k: sdG[k].detach().float().cpu() for k in keys_g})
    out.update({"D." + k: sdD[k].detach().float().cpu() for k in keys_d})
    return out


def _worker(rank, world, port, out_dir, dtype, bucket_mb, hr_scale, backend="gloo"):
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests"))
    local = rank if backend == "nccl" else 0  # RCCL: one rank per device; gloo: both ranks on cuda:0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(local), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from gan_sr_wind_field_amd import dist as wdist
    from oracle.gan import synthetic_batch

    torch.cuda.set_device(local)
    assert wdist.init_from_env(backend, single_rank=world == 1)
    assert dist.get_world_size() == world and dist.get_backend() == backend
    gan, cfg = _build_gan(dtype, local)
    dp = wdist.attach(gan, bucket_mb=bucket_mb, sync_bn=True)
    dp.stats.timing = True
    LR, HR, Z, x, y = synthetic_batch(2 * world, 16, 4, 4, seed=2001)
    HR = HR * hr_scale
    sl = slice(2 * rank, 2 * rank + 2)  # two samples per rank
    res = _two_iterations(gan, cfg, LR[sl], HR[sl], Z[sl], x, y)
    res["n_coll"] = dp.n_collectives
    # every element of both flat gradient buffers (and the classifier head's four tensors) went through a bucket, once
    res["grad_elems_expected"] = (gan.G.program().space.total + gan.D.features.program().space.total
                                  + sum(p.numel() for p in gan.D.classifier.parameters()))
    res["grad_elems_g"] = gan.G.program().space.total  # (a generator pass the loss guards discarded sent its buckets too)
    torch.cuda.synchronize()
    res["comm"] = dp.stats.summary(1)
    res["bn_layers"] = sum(1 for l in gan.D.features.program().layers if l.bn is not None)
    torch.save(res, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_mb,hr_scale", [(0.02, 1.0), (32.0, 1.0), (32.0, 1e-4)],
                         ids=["small_buckets", "default_buckets", "sr_normaliser_branch"])
def test_two_rank_step_equals_full_batch_hip(hip, tmp_path, bucket_mb, hr_scale):
    """``sr_normaliser_branch``: HR scaled by 1e-4 so that every physics normaliser is SR_max / 100 - the branch in
    which the maxima carry gradient (reference :773-814); under DP the fused loss path falls back to the composed
    ops and the gradient of the global maximum is routed to the rank that owns it (dist._GlobalMax)."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from oracle.gan import synthetic_batch

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), "fp32", bucket_mb, hr_scale), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    gan, cfg = _build_gan("fp32")
    LR, HR, Z, x, y = synthetic_batch(2 * world, 16, 4, 4, seed=2001)
    ref = _two_iterations(gan, cfg, LR, HR * hr_scale, Z, x, y)
    lr = cfg.training.learning_rate_d if hasattr(cfg.training, "learning_rate_d") else 1e-4
    from conftest import rel_l2
    for k, v in ref.items():
        assert torch.equal(r0[k], r1[k]), k  # replicas stay identical
        if k.startswith("grad"):
            # gradients BEFORE Adam: tight for the generator iteration (same weights on both sides: only the order of the
            # batch sums differs), the discriminator iteration's at the bound of the other fp32 D tests (its input was
            # generated by weights that already took one sign(g) step: see below)
            assert rel_l2(r0[k], v) < (1e-4 if k.startswith("gradG") else 2.5e-2), (k, rel_l2(r0[k], v))
            continue
        a, b = r0[k].numpy(), v.numpy()
        bad = np.abs(a - b) > 2e-6 + 5e-4 * np.abs(b)
        # Adam's first step moves every weight by lr * sign(g): an element whose gradient is at the rounding level of
        # the (differently ordered) batch sums may take the other sign - at most twice the step, on a handful of them
        assert bad.mean() <= 0.02 and (not bad.any() or np.abs(a - b)[bad].max() <= 2.05 * max(lr, 1e-4)), \
            (k, float(bad.mean()), float(np.abs(a - b).max()))
    assert r0["n_coll"] > (4 if bucket_mb < 1 else 2)  # gradient buckets of G and D + the classifier head
    # the communication ledger bench.py prints: SyncBN costs ONE collective per BatchNorm layer and pass for BOTH
    # inputs of the iteration (D(real) and D(fake) ride together) - forward all-gather + backward sum in the
    # D-iteration; the G-iteration runs D in eval mode (no batch statistics)
    comm = r0["comm"]
    assert comm["syncbn_collectives_per_step"] == 2 * r0["bn_layers"], comm
    assert comm["grad_bucket_collectives_per_step"] == r0["n_coll"]
    sent = r0["grad_elems_expected"] + int(comm["discarded_generator_passes_per_step"]) * r0["grad_elems_g"]
    assert abs(comm["grad_mbytes_per_step"] * 1e6 - 4 * sent) < 1e3, (comm, sent)
    assert comm["discarded_generator_passes_per_step"] == (1 if hr_scale < 1 else 0)  # SR normalisers: one repeated pass
    assert comm["scalar_collectives_per_step"] >= 2
    assert comm["timed"] and comm["exposed_grad_wait_ms_per_step"] >= 0.0


def test_single_rank_group_runs_the_data_parallel_step_over_rccl(hip, tmp_path):
    """A process group of ONE rank on backend "nccl": every collective of the data-parallel step - ReduceOp.AVG gradient
    buckets on the collective stream, all_gather_into_tensor SyncBN statistics, the backward sums, the loss scalars,
    the HIP-event ledger - goes through RCCL on the one-GPU box (the two-rank RCCL test below needs two devices), and
    the result equals the plain single-process step on the same two samples."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from oracle.gan import synthetic_batch

    world, port = 1, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), "fp32", 0.02, 1.0, "nccl"), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    gan, cfg = _build_gan("fp32")
    LR, HR, Z, x, y = synthetic_batch(2, 16, 4, 4, seed=2001)
    ref = _two_iterations(gan, cfg, LR, HR, Z, x, y)
    for k, v in ref.items():
        np.testing.assert_allclose(r0[k].numpy(), v.numpy(), rtol=5e-4, atol=2e-6, err_msg=k)
    comm = r0["comm"]
    assert comm["syncbn_collectives_per_step"] == 2 * r0["bn_layers"], comm
    assert comm["grad_bucket_collectives_per_step"] == r0["n_coll"] > 4
    assert comm["timed"] and comm["exposed_grad_wait_ms_per_step"] >= 0.0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X devices (RCCL refuses two ranks on one)")
def test_two_rank_step_equals_full_batch_rccl(hip, tmp_path):
    """the same equality over backend "nccl" (= RCCL over xGMI), one rank per device: ReduceOp.AVG buckets on the
    collective stream, all_gather_into_tensor SyncBN statistics.  Skipped on the one-GPU box."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from oracle.gan import synthetic_batch

    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), "fp32", 32.0, 1.0, "nccl"), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    gan, cfg = _build_gan("fp32")
    LR, HR, Z, x, y = synthetic_batch(2 * world, 16, 4, 4, seed=2001)
    ref = _two_iterations(gan, cfg, LR, HR, Z, x, y)
    from conftest import rel_l2
    for k, v in ref.items():
        assert torch.equal(r0[k], r1[k]), k
        if k.startswith("grad"):
            assert rel_l2(r0[k], v) < (1e-4 if k.startswith("gradG") else 2.5e-2), (k, rel_l2(r0[k], v))
        else:
            np.testing.assert_allclose(r0[k].numpy(), v.numpy(), rtol=5e-4, atol=2e-6, err_msg=k)
    assert r0["comm"]["syncbn_collectives_per_step"] == 2 * r0["bn_layers"]
    assert abs(r0["comm"]["grad_mbytes_per_step"] * 1e6 - 4 * r0["grad_elems_expected"]) < 1e3
