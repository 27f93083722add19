"""GPU parity of the HIP path at the widths, dtype and tile geometry the benchmark runs.

* the SHIPPED configuration (BASELINE.json configs[0], "C1": full 34.77 M-parameter G, bf 32 D with slicing,
  LR 16x16x10 -> HR 64x64x10) - eval outputs, one G-iteration and one D-iteration, fp32 and bf16, against the
  fixture recorded from the reference (tests/golden/c1_full_step.npz) and against the oracle on the host;
* the C3' tile geometry (16-level tiles of 4 x 8 x 16 voxels, several z tiles, single activation buffer for
  the 5x5x5 conv, 512-voxel 128-wide tiles, Z16 filter-gradient kernels) on a slab the oracle finishes in
  seconds: full-width G and D, bf16;
* the 6-iteration loss / weight trace of the reference, bf16.

Tolerances.  fp32: outputs 2e-5, losses 2e-4, G gradients (48 dense blocks deep) every tensor 2e-3, median 5e-4
and 95 % of them within 2e-4 + 1.5 x the reference's own fp32-vs-fp64 distance on that tensor (recorded in the
fixture; rel-L2); D gradients of the end-to-end D-iteration 2.5e-2 (ill-conditioned in the generated input at
batch 1: a 5e-6 perturbation of SR moves them by 1.2e-2, measured on the oracle).
bf16: outputs 2e-2; every other bound is DERIVED, per loss entry and per parameter tensor, from the distance
d_k between the fp32 oracle and the same oracle with bf16 *storage emulation* (``GSpec.bf16_storage``: every
tensor the MI355X path keeps in HBM rounded to bf16, fp32 accumulation): tol_k = floor + 2 d_k with floor
1e-2 (losses) / 2e-2 (gradients).  d_k is what ANY bf16-storage implementation of the reference's graph must
show; the HIP path has to stay within twice that of the fp32 truth.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import reload_wsr_env, REPO, rel_l2
from c1_case import LOSS_KEYS, c1_batch, c1_d_grads_fp64, c1_oracle_step, c1_specs, c1_states
from oracle import gan as ogan
from oracle import nets as onets

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"


def _report(name, payload):
    """measured distances -> gpurun_out/parity_<name>.json (DESIGN.md quotes them); never fails the test"""
    try:
        d = os.path.join(REPO, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, f"parity_{name}.json"), "w") as f:
            json.dump(payload, f, indent=1, sort_keys=True)
    except OSError:
        pass


def _gan(dtype, ini="local", **over):
    from gan_sr_wind_field_amd.config.config import Config
    from gan_sr_wind_field_amd.GAN_models.wind_field_GAN_3D import wind_field_GAN_3D
    import gan_sr_wind_field_amd

    cfg = Config(os.path.join(os.path.dirname(gan_sr_wind_field_amd.__file__), "config",
                              f"wind_field_GAN_3D_config_{ini}.ini"))
    cfg.is_train, cfg.is_test, cfg.is_use = True, False, False
    cfg.gpu_id, cfg.device = 0, torch.device(DEV)
    cfg.compute_dtype = dtype
    cfg.generator.dropout_probability = cfg.discriminator.dropout_probability = 0.0
    cfg.training.use_instance_noise = False
    cfg.training.niter = 150000
    cfg.training.d_g_train_period = 1
    for k, v in over.items():
        sec, key = k.split("__")
        setattr(getattr(cfg, sec), key, v)
    torch.manual_seed(2001)
    return wind_field_GAN_3D(cfg), cfg


def _bounds(truth: dict, emul: dict, floor: float, pooled: bool = False):
    """floor + 2 x emulated distance per tensor; ``pooled`` (discriminator gradients): at least the network's
    median emulated distance, see test_hip_networks._emulated_bf16_bounds"""
    from test_hip_networks import _emulated_bf16_bounds

    return _emulated_bf16_bounds(truth, emul, floor, pooled)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_c1_shipped_config_step(golden, hip, dtype):
    g = golden("c1_full_step.npz")
    r = c1_oracle_step(torch.float32, emulate_bf16=False)
    bf16 = dtype == "bf16"
    e = c1_oracle_step(torch.float32, emulate_bf16=True) if bf16 else None
    gan, cfg = _gan(dtype)
    assert sum(p.numel() for p in gan.G.parameters()) == 34769571
    sdG, sdD = c1_states()
    gan.G.load_state_dict(sdG)
    gan.D.load_state_dict(sdD)
    LR, HR, Z, x, y = (t.to(DEV) for t in c1_batch())
    gan.feed_xy_niter(x, y, torch.tensor(cfg.training.niter, device=DEV), 1, 1)
    rep = {}

    # ---- eval-mode outputs: SR field and D's logit on HR
    gan.G.eval()
    gan.D.eval()
    with torch.no_grad():
        sr = gan.G(LR, Z)
        d_hr = gan.D(HR)
    rep["sr_vs_reference"] = rel_l2(sr[:, :, ::2, ::2, :], T(g["sr_sub"]))
    rep["sr_vs_oracle"] = rel_l2(sr, r["sr"])
    rep["d_logit_rel"] = abs(float(d_hr) - float(g["d_hr_eval"])) / abs(float(g["d_hr_eval"]))
    assert rep["sr_vs_reference"] < (2e-2 if bf16 else 2e-5)
    assert rep["sr_vs_oracle"] < (2e-2 if bf16 else 2e-5)
    d_tol = 1e-2 + 2 * abs(float(e["d_hr_eval"]) - float(r["d_hr_eval"])) / abs(float(r["d_hr_eval"])) if bf16 else 2e-5
    assert rep["d_logit_rel"] < d_tol, (rep["d_logit_rel"], d_tol)

    # ---- G-iteration: the 8 loss entries (fp32 on both sides) and every parameter gradient
    gan.optimize_parameters(LR, HR, Z, 0)
    got = np.array([float(gan.get_G_train_loss_dict_ref()[k].detach()) for k in LOSS_KEYS])
    ref = np.asarray(g["G_losses"])
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-12)
    rep["G_losses_rel"] = dict(zip(LOSS_KEYS, rel.tolist()))
    if bf16:
        emul = np.asarray(e["G_losses"])
        tol = 1e-2 + 2 * np.abs(emul - np.asarray(r["G_losses"])) / np.maximum(np.abs(ref), 1e-12)
    else:
        tol = np.full(len(ref), 2e-4)
    assert (rel[ref != 0] < tol[ref != 0]).all(), (rep["G_losses_rel"], tol.tolist())
    assert (got[ref == 0] == 0).all()
    grads = {k: p.grad for k, p in gan.G.named_parameters()}
    errs = {k: rel_l2(grads[k], r["gG"][k]) for k in grads}
    # fp32, 48 dense blocks deep: the REFERENCE's own fp32 gradients sit a median 2.4e-4 (worst 9e-4) from an fp64
    # evaluation (recorded per tensor in the fixture: gG_floor) - rounding plus the odd LeakyReLU branch flip of
    # a near-zero activation, which moves one filter's gradient by ~1e-3.  So: every tensor < 2e-3, the median
    # < 5e-4, and at least 95 % of the tensors within 2e-4 + 1.5 x their recorded floor (measured: 293 of 297).
    floor = dict(zip((str(k) for k in g["gG_keys"]), g["gG_floor"]))
    lim = _bounds(r["gG"], e["gG"], 2e-2) if bf16 else {k: 2e-3 for k in grads}
    rep["gG_worst"] = sorted(((v, k, lim[k]) for k, v in errs.items()), reverse=True)[:5]
    rep["gG_median"] = float(np.median(list(errs.values())))
    bad = {k: (v, lim[k]) for k, v in errs.items() if not v < lim[k]}
    if not bf16:
        tight = sum(errs[k] < 2e-4 + 1.5 * floor[k] for k in errs)
        rep["gG_within_floor_bound"] = [int(tight), len(errs)]
    _report(f"c1_{dtype}", rep)
    assert not bad, bad
    if not bf16:
        assert rep["gG_median"] < 5e-4 and tight >= 0.95 * len(errs), (rep["gG_median"], tight)
    # the same gradients against the reference's own run: norm of every tensor + the recorded tensors
    l2 = np.array([float(grads[str(k)].double().norm()) for k in g["gG_keys"]])
    lim_n = np.array([lim[str(k)] for k in g["gG_keys"]])
    assert (np.abs(l2 / g["gG_l2"] - 1) < lim_n).all()
    for k in g.files:
        if k.startswith("gG."):
            assert rel_l2(grads[k[3:]], T(g[k])) < lim[k[3:]], k

    # ---- the Adam step just taken: the first step moves every weight by ~lr * sign(g), so a gradient element of
    # the wrong SIGN (only possible where it is ~0) shows as a 2 lr difference.  Few may, and nothing else.
    after = gan.G.state_dict()
    lr_g = cfg.training.learning_rate_g
    flips, total = 0, 0
    for k, w in r["sdG_after"].items():
        d = (after[k].cpu() - w).abs()
        assert float(d.max()) <= 2.001 * lr_g + 1e-7, k
        flips += int((d > 0.5 * lr_g).sum())
        total += d.numel()
    rep["adam_sign_flips_frac"] = flips / total
    assert flips / total < (0.05 if bf16 else 2e-3), flips / total
    # the D-iteration is compared on IDENTICAL generator weights (the oracle's post-step ones): a handful of
    # 2 lr weight differences would otherwise reach D's first-layer gradients at the 1e-2 level
    gan.G.load_state_dict(r["sdG_after"])

    # ---- D-iteration: loss, every parameter gradient, BatchNorm running statistics
    gan.optimize_parameters(LR, HR, Z, 1)
    d_loss = float(gan.get_D_loss_dict_ref()["train_loss"].detach())
    rep["D_loss_rel"] = abs(d_loss - float(g["D_loss"])) / abs(float(g["D_loss"]))
    dl_tol = 1e-2 + 2 * abs(e["D_loss"] - r["D_loss"]) / abs(r["D_loss"]) if bf16 else 2e-4
    assert rep["D_loss_rel"] < dl_tol, (rep["D_loss_rel"], dl_tol)
    gD = {k: p.grad for k, p in gan.D.named_parameters()}
    errs = {k: rel_l2(gD[k], r["gD"][k]) for k in gD}
    # fp32: these gradients are ILL-CONDITIONED in the fake input (train-mode BatchNorm over 80..10240 elements at
    # batch 1, ten layers deep): measured on the oracle, a 5e-6 relative perturbation of SR - the size of the fp32
    # difference between two generator implementations - moves them by up to 1.2e-2, and the oracle's own fp32
    # run is 4.4e-3 from fp64.  2.5e-2 here; the discriminator alone, on identical inputs, is held to the tight
    # per-tensor bounds in test_discriminator_fp32_vs_reference and the slab test below.
    lim = _bounds(r["gD"], e["gD"], 2e-2, pooled=True) if bf16 else {k: 2.5e-2 for k in gD}
    rep["gD_worst"] = sorted(((v, k, lim[k]) for k, v in errs.items()), reverse=True)[:5]
    bad = {k: (v, lim[k]) for k, v in errs.items() if not v < lim[k]}
    assert not bad, bad
    if not bf16:  # and within the same distance of an fp64 evaluation of the D-iteration
        d64 = c1_d_grads_fp64(r["sr_d"])
        assert max(rel_l2(gD[k].double().cpu(), d64[k]) for k in gD) < 2.5e-2
    for k in g.files:
        if k.startswith("gD."):
            assert rel_l2(gD[k[3:]], T(g[k])) < lim[k[3:]], k
    sd = gan.D.state_dict()
    for k, v in r["bn"].items():
        assert rel_l2(sd[k].float(), v) < (2e-2 if bf16 else 1e-5), k
    _report(f"c1_{dtype}", rep)


def test_c3_geometry_slab_generator_bf16(hip, monkeypatch):
    """Full-width G on LR 16x16x32 -> HR 64x64x32 with the kernels and tiles of the 128^3 benchmark: 16-level
    tiles (two z tiles everywhere), the single-buffer 5x5x5 144-wide kernel, 512-voxel 128-wide trunk tiles
    (WSR_CT_NOSMALL keeps this small volume off the 128-voxel variants), Z16 filter-gradient kernels.  Output
    and EVERY parameter gradient against the fp32 oracle, bounds from the bf16-storage emulation."""
    monkeypatch.setenv("WSR_CT_NOSMALL", "1")
    reload_wsr_env()
    from test_hip_networks import build_G

    spec = onets.GSpec()
    LR, HR, Z, x, y = ogan.synthetic_batch(1, 16, 32, 4, seed=77)
    gy = torch.randn((1, 3, 64, 64, 32), generator=torch.Generator().manual_seed(5))
    res = {}
    for mode in ("fp32", "emul"):
        sd = onets.deterministic_state(onets.g_param_shapes(spec), seed=111, scale=0.3)
        for v in sd.values():
            v.requires_grad_(True)
        sp = onets.GSpec(bf16_storage=mode == "emul")
        out = onets.generator_forward(sd, LR, Z, sp, training=False)
        (out * gy).sum().backward()
        res[mode] = (out.detach(), {k: v.grad for k, v in sd.items()})
    G, _ = build_G(spec, torch.bfloat16, 111, scale=0.3)
    G.eval()
    seen = []
    G.program().launch_probe = lambda tag, fn: (seen.append(tag.split(":")[0]), fn())
    out = G(LR.to(DEV), Z.to(DEV))
    (out * gy.to(DEV)).sum().backward()
    assert "fwd_dense_pre" in seen and "dgrad_dense0" in seen and "wgrad_tri" in seen
    e_out = rel_l2(out, res["fp32"][0])
    assert e_out < 2e-2, e_out
    lim = _bounds(res["fp32"][1], res["emul"][1], 2e-2)
    errs = {k: rel_l2(p.grad, res["fp32"][1][k]) for k, p in G.named_parameters()}
    _report("c3_slab_G_bf16", {"out": e_out, "emul_out": rel_l2(res["emul"][0], res["fp32"][0]),
                               "worst": sorted(((v, k, lim[k]) for k, v in errs.items()), reverse=True)[:8],
                               "median": float(np.median(list(errs.values())))})
    bad = {k: (v, lim[k]) for k, v in errs.items() if not v < lim[k]}
    assert not bad, bad


@pytest.mark.parametrize("train", [False, True], ids=["eval_input_grad", "train_param_grads"])
def test_c3_geometry_slab_discriminator_bf16(hip, train):
    """Full-width D (bf 32, no slicing) on 128x128x32 inputs - more than 19 levels, so block 0 halves z as at
    128^3 (reference Discriminator_3D.py:75) and the strided (4,4,3) convs take their 16-level tiles.
    eval mode (the G-iteration's use): logit and input gradient; train mode: logit and every parameter gradient."""
    from test_hip_networks import build_D

    spec = onets.DSpec(bf=32, nz=32)
    gen = torch.Generator().manual_seed(9)
    x = torch.rand((1, 3, 128, 128, 32), generator=gen) * 2 - 1
    res = {}
    for mode in ("fp32", "emul"):
        sd = onets.deterministic_state(onets.d_param_shapes(spec), seed=121, scale=1.0)
        params = {k: v for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
        for v in params.values():
            v.requires_grad_(train)
        xr = x.clone().requires_grad_(not train)
        sp = onets.DSpec(bf=32, nz=32, bf16_storage=mode == "emul")
        out = onets.discriminator_forward(sd, xr, sp, training=train)
        out.sum().backward()
        res[mode] = (float(out), {k: v.grad for k, v in params.items()} if train else {"x": xr.grad})
    D, _ = build_D(spec, torch.bfloat16, 121)
    D.train(train)
    for p in D.parameters():
        p.requires_grad = train
    xd = x.to(DEV).requires_grad_(not train)
    out = D(xd)
    out.sum().backward()
    tol = 1e-2 + 2 * abs(res["emul"][0] - res["fp32"][0]) / abs(res["fp32"][0])
    assert abs(float(out) - res["fp32"][0]) / abs(res["fp32"][0]) < tol
    got = {k: p.grad for k, p in D.named_parameters()} if train else {"x": xd.grad}
    lim = _bounds(res["fp32"][1], res["emul"][1], 2e-2, pooled=True)
    errs = {k: rel_l2(got[k], res["fp32"][1][k]) for k in got}
    _report(f"c3_slab_D_bf16_{'train' if train else 'eval'}",
            {"worst": sorted(((v, k, lim[k]) for k, v in errs.items()), reverse=True)[:6]})
    bad = {k: (v, lim[k]) for k, v in errs.items() if not v < lim[k]}
    assert not bad, bad


def test_gan_train_step_trace_bf16_vs_reference(golden, hip):
    """6 iterations (G, D, D, G, G, D) in bf16 against the reference's fp32 trace of the same widths
    (gan_trace_plain_w8.npz): every loss entry within 1e-2 + 3x the largest deviation the bf16-storage emulation of
    the oracle shows on that entry over the trace (measured on the host, same run; the relativistic adversarial
    term is a difference of logits and moves by ~1e-2 under bf16, the batch-of-2 train-mode D loss by ~3e-2),
    Adam-updated weight sums within 1e-3."""
    g = golden("gan_trace_plain_w8.npz")
    keys = LOSS_KEYS

    def nets_and_batch():
        gs = onets.GSpec(in_channels=4, nf=16, n_rrdb=1, gc=8, tf=8, hr_kern=5, upscale=4)
        ds = onets.DSpec(bf=8, nz=4, enable_slicing=True)
        sdG = onets.deterministic_state(onets.g_param_shapes(gs), seed=41, scale=0.5)
        sdD = onets.deterministic_state(onets.d_param_shapes(ds), seed=43, scale=1.0)
        return gs, ds, sdG, sdD, ogan.synthetic_batch(2, 16, 4, 4, seed=2001)

    # emulated deviations, per iteration and entry
    gs, ds, sdG, sdD, (LR, HR, Z, x, y) = nets_and_batch()
    gs.bf16_storage = ds.bf16_storage = True
    em = ogan.OracleGAN(sdG, sdD, gs, ds, ogan.TrainSpec(use_instance_noise=False, d_g_train_period=2))
    em.feed_xy(x, y)
    dev_G, dev_D = {}, {}
    for row, it in enumerate(g["its"]):
        kind = em.optimize_parameters(LR, HR, Z, int(it))
        if int(it) > 4:
            em.update_learning_rate()
        if kind == "G":
            ref = g["G_losses"][row]
            dev_G[row] = np.abs(np.array([float(em.G_losses[k]) for k in keys]) - ref) / np.maximum(np.abs(ref), 1e-12)
        else:
            dev_D[row] = abs(float(em.D_loss) - g["D_loss"][row]) / abs(g["D_loss"][row])

    gan, cfg = _gan("bf16", generator__num_features=16, generator__num_RRDB=1, generator__RDB_growth_chan=8,
                    generator__terrain_number_of_features=8, discriminator__num_features=8,
                    gan_config__number_of_z_layers=4, training__d_g_train_period=2)
    gs, ds, sdG, sdD, batch = nets_and_batch()
    gan.G.load_state_dict(sdG)
    gan.D.load_state_dict(sdD)
    LR, HR, Z, x, y = (t.to(DEV) for t in batch)
    gan.feed_xy_niter(x, y, torch.tensor(cfg.training.niter, device=DEV), 1, 2)
    rep = {}
    for row, it in enumerate(g["its"]):
        gan.optimize_parameters(LR, HR, Z, int(it))
        if int(it) > 4:
            gan.update_learning_rate()
        if g["kinds"][row]:
            ref = g["G_losses"][row]
            got = np.array([float(gan.get_G_train_loss_dict_ref()[k].detach()) for k in keys])
            rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-12)
            # one scalar's emulated deviation is a single draw: use the largest one that entry showed over the trace
            tol = 1e-2 + 3 * np.max(np.stack(list(dev_G.values())), axis=0)
            rep[f"it{it}_G"] = rel.tolist()
            assert (rel[ref != 0] < tol[ref != 0]).all(), (int(it), rel.tolist(), tol.tolist())
        else:
            rel = abs(float(gan.get_D_loss_dict_ref()["train_loss"].detach()) - g["D_loss"][row]) / abs(g["D_loss"][row])
            rep[f"it{it}_D"] = rel
            assert rel < 1e-2 + 3 * max(dev_D.values()), (int(it), rel, dev_D)
        sG, sD = gan.G.state_dict(), gan.D.state_dict()
        wg = [float(sG[k].double().abs().sum())
              for k in ("model.0.0.weight", "hr_convs.2.weight", "model.1.module.0.RDBs.1.LFF.bias")]
        wd = [float(sD[k].double().abs().sum())
              for k in ("features.0.0.0.weight", "classifier.2.weight", "features.1.1.1.running_var")]
        np.testing.assert_allclose(wg, g["wsum_g"][row], rtol=1e-3, err_msg=f"it={it}")
        np.testing.assert_allclose(wd, g["wsum_d"][row], rtol=1e-3, err_msg=f"it={it}")
    for k in g.files:
        if k.startswith("final_G."):
            assert rel_l2(gan.G.state_dict()[k[8:]], T(g[k])) < 2e-3, k
        if k.startswith("final_D."):  # (running statistics of bf16-stored activations: 2e-2)
            assert rel_l2(gan.D.state_dict()[k[8:]], T(g[k])) < (2e-2 if "running_" in k else 2e-3), k
    _report("trace_bf16", rep)


_c3_oracle = {}


def _c3_full_oracle():
    """oracle forward at the HEADLINE size, once per session: LR 32x32x128 -> SR 128^3 through the full 34.77 M-parameter
    G (21.6 TFLOP: 15-40 s on the box's host cores), and D's eval logit on HR at 128^3 (23.4 M parameters)"""
    if not _c3_oracle:
        import time
        gs, ds = onets.GSpec(), onets.DSpec(bf=32, nz=128, enable_slicing=False)
        sdG = onets.deterministic_state(onets.g_param_shapes(gs), seed=211, scale=0.3)
        sdD = onets.deterministic_state(onets.d_param_shapes(ds), seed=213, scale=1.0)
        LR, HR, Z, x, y = ogan.synthetic_batch(1, 32, 128, 4, seed=2001)
        t0 = time.time()
        with torch.no_grad():
            sr = onets.generator_forward(sdG, LR, Z, gs, training=False)
            d_hr = onets.discriminator_forward(sdD, HR, ds, training=False)
            ds16 = onets.DSpec(bf=32, nz=128, enable_slicing=False, bf16_storage=True)
            d_em = onets.discriminator_forward(sdD, HR, ds16, training=False)
        _c3_oracle.update(gs=gs, ds=ds, sdG=sdG, sdD=sdD, LR=LR, HR=HR, Z=Z, sr=sr, d_hr=d_hr, d_em=d_em,
                          host_s=time.time() - t0)
    return _c3_oracle


@pytest.mark.timeout(900)
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_c3_full_size_forward_vs_oracle(hip, dtype):
    """BASELINE's headline shape against the ORACLE at full size (not a slab, not a property): the SR field of the
    full-width generator on LR 32x32x128 -> 128^3 and the discriminator's eval logit on a 128^3 HR volume, both HIP
    programs.  fp32: rel-L2 2e-5 (SR) / 2e-5 of the logit; bf16: 2e-2 (SR) and the logit within 1e-2 + twice the
    distance the oracle's bf16-storage emulation shows."""
    from test_hip_networks import build_D, build_G

    o = _c3_full_oracle()
    dt = torch.float32 if dtype == "fp32" else torch.bfloat16
    from gan_sr_wind_field_amd.CNN_models.Generator_3D_Resnet_ESRGAN import Generator_3D
    from gan_sr_wind_field_amd.CNN_models.Discriminator_3D import Discriminator_3D
    gs, ds = o["gs"], o["ds"]
    G = Generator_3D(gs.in_channels, gs.out_channels, gs.nf, gs.n_rrdb, upscale=gs.upscale, hr_kern_size=gs.hr_kern,
                     RDB_gc=gs.gc, terrain_number_of_features=gs.tf, dropout_probability=gs.dropout_p,
                     use_mixed_precision=dt == torch.bfloat16)
    G.load_state_dict(o["sdG"])
    G = G.to(DEV).eval()
    D = Discriminator_3D(ds.in_channels, ds.bf, feat_kern_size=ds.feat_kern, number_of_z_layers=ds.nz,
                         enable_slicing=False, dropout_probability=ds.dropout_p, use_mixed_precision=dt == torch.bfloat16)
    D.load_state_dict(o["sdD"])
    D = D.to(DEV).eval()
    with torch.no_grad():
        sr = G(o["LR"].to(DEV), o["Z"].to(DEV))
        d_hr = D(o["HR"].to(DEV))
    assert sr.shape == (1, 3, 128, 128, 128)
    e_sr = rel_l2(sr, o["sr"])
    ref = float(o["d_hr"])
    e_d = abs(float(d_hr) - ref) / abs(ref)
    d_tol = 2e-5 if dtype == "fp32" else 1e-2 + 2 * abs(float(o["d_em"]) - ref) / abs(ref)
    _report(f"c3_full_forward_{dtype}", {"sr_vs_oracle": e_sr, "d_logit_rel": e_d, "d_logit_tol": d_tol,
                                         "oracle_host_s": o["host_s"], "d_logit": ref})
    assert e_sr < (2e-5 if dtype == "fp32" else 2e-2), e_sr
    assert e_d < d_tol, (e_d, d_tol)
    del G, D
    torch.cuda.empty_cache()


@pytest.mark.parametrize("case", ["plain", "physics_nonfinite", "total_nonfinite", "sr_normaliser"])
def test_generator_iteration_guards_vs_oracle(hip, case):
    """The reference's non-finite guards of a generator iteration (wind_field_GAN_3D.py:434-460) on the HIP path, fp32,
    against the oracle on the same batch and weights:

    * ``physics_nonfinite`` - two equal z levels in Z: the terrain-following derivative divides by zero, the four physics
      terms are dropped from the total and the step runs on adversarial + pixel loss alone;
    * ``total_nonfinite`` - an infinite HR value: the total is not finite, backward still runs, the Adam step is skipped
      (weights and optimizer state untouched);
    * ``sr_normaliser`` - a generator whose output is far more than 100 x HR: the gradient normalisers come from SR and
      the reference differentiates through their maxima (the fused loss kernels hand over to the composed ops).

    Loss entries (finite ones rtol 1e-3, the others by class), gradients of three generator tensors, and whether the
    step was taken."""
    gs = onets.GSpec(in_channels=4, nf=16, n_rrdb=1, gc=8, tf=8, hr_kern=5, upscale=4)
    ds = onets.DSpec(bf=8, nz=4, enable_slicing=True)
    sdG = onets.deterministic_state(onets.g_param_shapes(gs), seed=41, scale=0.5)
    sdD = onets.deterministic_state(onets.d_param_shapes(ds), seed=43, scale=1.0)
    LR, HR, Z, x, y = ogan.synthetic_batch(2, 16, 4, 4, seed=2001)
    if case == "physics_nonfinite":
        Z[..., 2] = Z[..., 1]
    elif case == "total_nonfinite":
        HR[0, 1, 3, 4, 1] = float("inf")
    elif case == "sr_normaliser":
        sdG["hr_convs.2.weight"] = sdG["hr_convs.2.weight"] * 3e4
    gan, cfg = _gan("fp32", generator__num_features=16, generator__num_RRDB=1, generator__RDB_growth_chan=8,
                    generator__terrain_number_of_features=8, discriminator__num_features=8,
                    gan_config__number_of_z_layers=4, training__d_g_train_period=1)
    gan.G.load_state_dict(sdG)
    gan.D.load_state_dict(sdD)
    gan.feed_xy_niter(x.to(DEV), y.to(DEV), torch.tensor(cfg.training.niter, device=DEV), 1, 1)
    w0 = {k: v.detach().clone() for k, v in gan.G.state_dict().items()}
    it = 2  # (period 1: even iterations are generator iterations)
    gan.optimize_parameters(LR.to(DEV), HR.to(DEV), Z.to(DEV), it)

    ref = ogan.OracleGAN({k: v.clone() for k, v in sdG.items()}, {k: v.clone() for k, v in sdD.items()}, gs, ds,
                         ogan.TrainSpec(use_instance_noise=False, d_g_train_period=1))
    ref.feed_xy(x, y)
    assert ref.optimize_parameters(LR, HR, Z, it) == "G"

    got = {k: float(gan.get_G_train_loss_dict_ref()[k].detach()) for k in LOSS_KEYS}
    want = {k: float(ref.G_losses[k]) for k in LOSS_KEYS}
    for k in LOSS_KEYS:
        if np.isfinite(want[k]):
            assert got[k] == pytest.approx(want[k], rel=1e-3, abs=1e-7), (case, k, got, want)
        else:
            assert not np.isfinite(got[k]), (case, k, got, want)
    if case == "physics_nonfinite":
        assert not all(np.isfinite(want[k]) for k in ("z_gradient", "divergence")) and np.isfinite(want["total"])
    stepped = bool(np.isfinite(want["total"]))
    assert stepped == (case != "total_nonfinite")
    names = dict(gan.G.named_parameters())
    for k in ("model.0.0.weight", "hr_convs.2.weight", "model.1.module.0.RDBs.1.LFF.bias"):
        if stepped:  # the gradients that drove the step
            assert rel_l2(names[k].grad.cpu(), ref.sdG[k].grad) < 2e-3, (case, k)
            assert not torch.equal(names[k].detach(), w0[k])
        else:
            assert torch.equal(names[k].detach(), w0[k]), (case, k)
    st = gan.optimizer_G.state_dict()["state"]
    assert (len(st) > 0 and float(st[0]["step"]) == 1.0) if stepped else all(float(s["step"]) == 0.0 for s in st.values())


def test_speculative_guards_back_off_when_they_keep_firing(hip, monkeypatch):
    """A batch whose physics terms are non-finite EVERY iteration (two equal z levels): the speculative generator pass
    (flags read behind the backward pass) is discarded and repeated - after SPEC_MISS_LIMIT consecutive misses the model
    stops speculating for SPEC_BACKOFF generator iterations (no more discarded passes), says so ONCE in the status log,
    and every iteration's losses still equal the oracle's."""
    from gan_sr_wind_field_amd.GAN_models import wind_field_GAN_3D as mod

    monkeypatch.setattr(mod, "SPEC_MISS_LIMIT", 2)
    monkeypatch.setattr(mod, "SPEC_BACKOFF", 3)
    gs = onets.GSpec(in_channels=4, nf=16, n_rrdb=1, gc=8, tf=8, hr_kern=5, upscale=4)
    ds = onets.DSpec(bf=8, nz=4, enable_slicing=True)
    sdG = onets.deterministic_state(onets.g_param_shapes(gs), seed=41, scale=0.5)
    sdD = onets.deterministic_state(onets.d_param_shapes(ds), seed=43, scale=1.0)
    LR, HR, Z, x, y = ogan.synthetic_batch(2, 16, 4, 4, seed=2001)
    Z[..., 2] = Z[..., 1]
    gan, cfg = _gan("fp32", generator__num_features=16, generator__num_RRDB=1, generator__RDB_growth_chan=8,
                    generator__terrain_number_of_features=8, discriminator__num_features=8,
                    gan_config__number_of_z_layers=4, training__d_g_train_period=1)
    gan.G.load_state_dict(sdG)
    gan.D.load_state_dict(sdD)
    gan.get_new_status_logs()
    gan.feed_xy_niter(x.to(DEV), y.to(DEV), torch.tensor(cfg.training.niter, device=DEV), 1, 1)
    ref = ogan.OracleGAN({k: v.clone() for k, v in sdG.items()}, {k: v.clone() for k, v in sdD.items()}, gs, ds,
                         ogan.TrainSpec(use_instance_noise=False, d_g_train_period=1))
    ref.feed_xy(x, y)
    dev = [t.to(DEV) for t in (LR, HR, Z)]
    retries = []
    for it in range(2, 18, 2):  # eight generator iterations
        gan.optimize_parameters(*dev, it)
        assert ref.optimize_parameters(LR, HR, Z, it) == "G"
        retries.append(getattr(gan, "spec_retries", 0))
        for k in LOSS_KEYS:
            want = float(ref.G_losses[k])
            got = float(gan.get_G_train_loss_dict_ref()[k].detach())
            assert (got == pytest.approx(want, rel=2e-3, abs=1e-7)) if np.isfinite(want) else not np.isfinite(got), (it, k)
    # misses at iterations 1, 2 -> three careful iterations -> misses at 6, 7 -> careful again
    assert retries == [1, 2, 2, 2, 2, 3, 4, 4], retries
    logs = [line for line in gan.get_new_status_logs() if "guards fired" in line]
    assert len(logs) == 1, logs
    assert rel_l2(dict(gan.G.named_parameters())["hr_convs.2.weight"].detach().cpu(), ref.sdG["hr_convs.2.weight"].detach()) < 5e-3


@pytest.mark.parametrize("gan_type,pix", [("relativistic", "l2"), ("relativisticavg", "l2"), ("relativistic", "l1")])
def test_loss_variants_vs_oracle(hip, gan_type, pix):
    """The configuration branches the shipped ini does not take - ``gan_type = relativistic`` (reference
    wind_field_GAN_3D.py:358-359, 545-547: no batch means in the adversarial terms) and ``pixel_criterion = l2`` (the
    squared-error sum of the fused content-loss pass) - one discriminator and one generator iteration in fp32 against
    the oracle: loss entries rtol 1e-3, gradients of three tensors of each network."""
    gs = onets.GSpec(in_channels=4, nf=16, n_rrdb=1, gc=8, tf=8, hr_kern=5, upscale=4)
    ds = onets.DSpec(bf=8, nz=4, enable_slicing=True)
    sdG = onets.deterministic_state(onets.g_param_shapes(gs), seed=41, scale=0.5)
    sdD = onets.deterministic_state(onets.d_param_shapes(ds), seed=43, scale=1.0)
    LR, HR, Z, x, y = ogan.synthetic_batch(2, 16, 4, 4, seed=2001)
    gan, cfg = _gan("fp32", generator__num_features=16, generator__num_RRDB=1, generator__RDB_growth_chan=8,
                    generator__terrain_number_of_features=8, discriminator__num_features=8,
                    gan_config__number_of_z_layers=4, training__d_g_train_period=1, training__gan_type=gan_type,
                    training__pixel_criterion=pix)
    gan.G.load_state_dict(sdG)
    gan.D.load_state_dict(sdD)
    gan.feed_xy_niter(x.to(DEV), y.to(DEV), torch.tensor(cfg.training.niter, device=DEV), 1, 1)
    ref = ogan.OracleGAN({k: v.clone() for k, v in sdG.items()}, {k: v.clone() for k, v in sdD.items()}, gs, ds,
                         ogan.TrainSpec(use_instance_noise=False, d_g_train_period=1, gan_type=gan_type, pixel_criterion=pix))
    ref.feed_xy(x, y)
    dev = [t.to(DEV) for t in (LR, HR, Z)]
    for it in (1, 2):
        gan.optimize_parameters(*dev, it)
        kind = ref.optimize_parameters(LR, HR, Z, it)
        if kind == "D":
            assert float(gan.get_D_loss_dict_ref()["train_loss"].detach()) == pytest.approx(float(ref.D_loss), rel=1e-3)
            names = dict(gan.D.named_parameters())
            for k in ("features.0.0.0.weight", "features.1.1.0.weight", "classifier.2.weight"):
                # (train-mode BatchNorm at batch 2 amplifies fp32 rounding: the bound of the other fp32 D tests)
                assert rel_l2(names[k].grad.cpu(), ref.sdD[k].grad) < 2.5e-2, (it, k)
        else:
            for k in LOSS_KEYS:
                assert float(gan.get_G_train_loss_dict_ref()[k].detach()) == pytest.approx(float(ref.G_losses[k]), rel=1e-3,
                                                                                          abs=1e-7), (it, k)
            names = dict(gan.G.named_parameters())
            for k in ("model.0.0.weight", "hr_convs.2.weight", "model.1.module.0.RDBs.1.LFF.bias"):
                assert rel_l2(names[k].grad.cpu(), ref.sdG[k].grad) < 2e-3, (it, k)


def test_checkpoint_resume_equals_uninterrupted_run(hip, tmp_path):
    """``save_model`` after four iterations, ``load_model`` into a fresh model, two more iterations: bit for bit the
    weights of the run that was never interrupted (deterministic kernels; the one-launch Adam keeps its step counts on
    the host and writes them into the state when a checkpoint asks) - and the optimizer state in the file is what
    ``torch.optim.Adam`` itself loads (the reference's ``load_model``, baseGAN.py:39-63)."""
    over = dict(generator__num_features=16, generator__num_RRDB=1, generator__RDB_growth_chan=8,
                generator__terrain_number_of_features=8, discriminator__num_features=8,
                gan_config__number_of_z_layers=4, training__d_g_train_period=1)
    LR, HR, Z, x, y = (t.to(DEV) for t in ogan.synthetic_batch(2, 16, 4, 4, seed=2001))

    def fresh():
        gan, cfg = _gan("bf16", **over)
        cfg.env.this_runs_folder = str(tmp_path)
        gan.feed_xy_niter(x, y, torch.tensor(cfg.training.niter, device=DEV), 1, 1)
        return gan

    a = fresh()
    for it in (1, 2, 3, 4):
        a.optimize_parameters(LR, HR, Z, it)
    a.save_model(str(tmp_path), 0, 4)
    for it in (5, 6):
        a.optimize_parameters(LR, HR, Z, it)

    b = fresh()
    assert b.load_model(str(tmp_path / "G_4.pth"), str(tmp_path / "D_4.pth"), str(tmp_path / "state_4.pth")) == (0, 4)
    b.G.train()
    for it in (5, 6):
        b.optimize_parameters(LR, HR, Z, it)
    for net in ("G", "D"):
        sa, sb = getattr(a, net).state_dict(), getattr(b, net).state_dict()
        for k in sa:
            assert torch.equal(sa[k], sb[k]), (net, k)
    state = torch.load(str(tmp_path / "state_4.pth"), map_location="cpu")
    for sd_opt, net in zip(state["optimizers"], (a.G, a.D)):
        assert all(float(s["step"]) == 2.0 for s in sd_opt["state"].values())  # two G- and two D-iterations
        plain = torch.optim.Adam([torch.nn.Parameter(p.detach().cpu().clone()) for p in net.parameters()])
        plain.load_state_dict(sd_opt)
        assert len(plain.state) == len(sd_opt["state"]) > 0
