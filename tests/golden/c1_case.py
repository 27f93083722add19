"""The shipped-configuration ("C1", BASELINE.json configs[0]) train-step case shared by the fixture generator
(make_golden.py runs the REAL reference on it), the oracle test and the GPU parity tests: full-width G and D,
LR 16x16x10 -> HR 64x64x10, batch 1, dropout and instance noise off, one G-iteration then one D-iteration."""
import torch

from oracle import gan as ogan
from oracle import nets as onets

G_SEED, G_SCALE, D_SEED, D_SCALE = 101, 0.3, 103, 1.0
LOSS_KEYS = ["total", "adversarial", "pix", "xy_gradient", "z_gradient", "divergence", "xy_divergence", "feature_D"]


def c1_specs(n_rrdb: int = 16):
    return onets.GSpec(n_rrdb=n_rrdb), onets.DSpec(bf=32, nz=10, enable_slicing=True)


def c1_states(dtype=torch.float32, n_rrdb: int = 16):
    gs, ds = c1_specs(n_rrdb)
    sdG = onets.deterministic_state(onets.g_param_shapes(gs), seed=G_SEED, scale=G_SCALE)
    sdD = onets.deterministic_state(onets.d_param_shapes(ds), seed=D_SEED, scale=D_SCALE)
    if dtype != torch.float32:
        sdG = {k: v.to(dtype) for k, v in sdG.items()}
        sdD = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sdD.items()}
    return sdG, sdD


def c1_batch(dtype=torch.float32):
    return tuple(t.to(dtype) for t in ogan.synthetic_batch(1, 16, 10, 4, seed=2001))


_cache = {}


def c1_oracle_step(dtype=torch.float32, emulate_bf16: bool = False):
    """-> dict(sr, d_hr_eval, G_losses, gG, D_loss, gD, bn) of the oracle (cached per argument pair)"""
    key = (dtype, emulate_bf16)
    if key in _cache:
        return _cache[key]
    gs, ds = c1_specs()
    gs.bf16_storage = ds.bf16_storage = emulate_bf16
    sdG, sdD = c1_states(dtype)
    LR, HR, Z, x, y = c1_batch(dtype)
    with torch.no_grad():
        sr = onets.generator_forward(sdG, LR, Z, gs, training=False)
        d_hr = onets.discriminator_forward(sdD, HR, ds, training=False)
    ref = ogan.OracleGAN(sdG, sdD, gs, ds, ogan.TrainSpec(use_instance_noise=False, d_g_train_period=1))
    if dtype != torch.float32:
        ref.niter = ref.niter.to(dtype)
    ref.feed_xy(x, y)
    assert ref.optimize_parameters(LR, HR, Z, 0) == "G"
    out = {"sr": sr, "d_hr_eval": d_hr, "G_losses": [float(ref.G_losses[k]) for k in LOSS_KEYS],
           "gG": {k: v.grad.detach().clone() for k, v in sdG.items()}}
    with torch.no_grad():  # the SR field the D-iteration will see (G after its Adam step, eval mode)
        out["sr_d"] = onets.generator_forward(sdG, LR, Z, gs, training=False)
    out["sdG_after"] = {k: v.detach().clone() for k, v in sdG.items()}
    assert ref.optimize_parameters(LR, HR, Z, 1) == "D"
    out["D_loss"] = float(ref.D_loss)
    out["gD"] = {k: v.grad.detach().clone() for k, v in sdD.items() if v.is_floating_point() and v.grad is not None}
    out["bn"] = {k: v.detach().clone() for k, v in sdD.items() if "running_" in k}
    _cache[key] = out
    return out


def c1_d_grads_fp64(sr_d: torch.Tensor):
    """fp64 evaluation of the D-iteration's gradients for a given SR field: the conditioning yard-stick of the
    fp32 tolerances (10 train-mode BatchNorm backward stages at batch 1)."""
    key = ("d64", float(sr_d.double().abs().sum()))
    if key in _cache:
        return _cache[key]
    _, ds = c1_specs()
    _, sdD = c1_states(torch.float64)
    _, HR, _, _, _ = c1_batch(torch.float64)
    ref = ogan.OracleGAN({"unused": torch.zeros(1, dtype=torch.float64)}, sdD, None, ds, ogan.TrainSpec(use_instance_noise=False, d_g_train_period=1))
    ref.niter = ref.niter.double()
    it = torch.tensor(1)
    ref.make_labels(it, 1)
    ref.HR_labels, ref.fake_labels = ref.HR_labels.double(), ref.fake_labels.double()
    y_real, y_fake = ref.D_pair(HR, sr_d.double(), it, train_D=True)
    ref.D_loss_value(y_real, y_fake).backward()
    _cache[key] = {k: v.grad.detach().clone() for k, v in sdD.items() if v.is_floating_point() and v.grad is not None}
    return _cache[key]
