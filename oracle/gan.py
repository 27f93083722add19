"""Oracle restatement of the wind_field_GAN_3D train step (G-/D-iteration).

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.

Restates GAN_models/wind_field_GAN_3D.py:207-304,342-474,476-500,532-678 and
tools/trainingtricks.py:18-59 on the CPU.  RNG calls are issued in the same
order as the reference so a seeded run reproduces its loss trace (pinned by
``tests/golden/gan_trace_*.npz``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import nets, physics

Tensor = torch.Tensor


@dataclass
class TrainSpec:
    """[TRAINING] section of the ini (config/config.py:213-289)."""

    lr_g: float = 8e-5
    lr_d: float = 8e-5
    wd_g: float = 0.0
    wd_d: float = 0.0
    beta1_g: float = 0.9
    beta1_d: float = 0.9
    lr_steps: List[int] = field(default_factory=lambda: [10000, 30000, 50000, 70000, 100000])
    lr_gamma: float = 0.5
    gan_type: str = "relativisticavg"
    w_adv: float = 0.0005
    w_pix: float = 0.136
    w_gxy: float = 3.064
    w_gz: float = 0.0
    w_div: float = 0.366
    w_divxy: float = 0.721
    pixel_criterion: Optional[str] = "l1"
    d_g_train_ratio: int = 1
    d_g_train_period: int = 50
    use_noisy_labels: bool = False
    use_one_sided_label_smoothing: bool = True
    flip_labels: bool = False
    use_instance_noise: bool = True
    niter: int = 150000
    scale: int = 4


def noisy_labels(is_real: bool, batch: int, std: float, false_val: Tensor, true_val: Tensor) -> Tensor:
    """trainingtricks.py:18-46: N(0, std) drawn on the CPU even when std == 0."""
    v = torch.normal(mean=0.0, std=torch.full(torch.Size([batch]), std))
    v = v + (true_val if is_real else false_val)
    return v.clamp(0.0, 1.0)


def instance_noise(sigma_base: float, shape, it: Tensor, niter: Tensor) -> Tensor:
    """trainingtricks.py:49-59: *uniform* [0,1) noise times sqrt(sigma(1-(it-1)/niter))."""
    noise = torch.rand(shape)
    return noise * torch.sqrt(torch.tensor(sigma_base) * (1 - (it - 1) / niter))


class OracleGAN:
    """Functional mirror of ``wind_field_GAN_3D`` on explicit state dicts."""

    def __init__(self, sdG: Dict[str, Tensor], sdD: Dict[str, Tensor], gs: nets.GSpec,
                 ds: nets.DSpec, ts: TrainSpec):
        self.gs, self.ds, self.ts = gs, ds, ts
        self.sdG, self.sdD = sdG, sdD
        self.pG = [v for k, v in sdG.items()]
        self.pD = [v for k, v in sdD.items() if v.is_floating_point() and "running_" not in k]
        for p in self.pG + self.pD:
            p.requires_grad_(True)
        # wind_field_GAN_3D.py:151-174
        self.opt_G = torch.optim.Adam(self.pG, lr=ts.lr_g, weight_decay=ts.wd_g, betas=(ts.beta1_g, 0.999))
        self.opt_D = torch.optim.Adam(self.pD, lr=ts.lr_d, weight_decay=ts.wd_d, betas=(ts.beta1_d, 0.999))
        self.sched = []
        if ts.lr_steps:
            self.sched = [
                torch.optim.lr_scheduler.MultiStepLR(o, ts.lr_steps, gamma=ts.lr_gamma)
                for o in (self.opt_G, self.opt_D)
            ]
        self._first_d_forward = True
        self.G_losses: Dict[str, Tensor] = {}
        self.D_loss: Optional[Tensor] = None
        self.niter = torch.tensor(ts.niter)

    # -- inputs -------------------------------------------------------------
    def feed_xy(self, x: Tensor, y: Tensor):
        self.x, self.y = x, y

    # -- labels: wind_field_GAN_3D.py:627-678 ---------------------------------
    def make_labels(self, it: Tensor, batch: int):
        ts = self.ts
        real, fake = torch.tensor(1.0), torch.tensor(0.0)
        if ts.use_one_sided_label_smoothing and ts.flip_labels:
            fake = torch.tensor(0.1) - 0.1 * it / self.niter
        elif ts.use_one_sided_label_smoothing:
            real = torch.tensor(0.9) + 0.1 * it / self.niter
        std = 0.05 if ts.use_noisy_labels else 0.0
        pred_real, pred_fake = (False, True) if ts.flip_labels else (True, False)
        self.HR_labels = noisy_labels(pred_real, batch, std, fake, real).squeeze()
        self.fake_labels = noisy_labels(pred_fake, batch, std, fake, real).squeeze()

    # -- D forward pair: wind_field_GAN_3D.py:221-304 ------------------------
    def D_pair(self, HR: Tensor, SR: Tensor, it: Tensor, train_D: bool):
        if self._first_d_forward:  # the device_check string draws one noise tensor (:228-246)
            instance_noise(2.0, HR.shape, it, self.niter)
            self._first_d_forward = False
        sigma = 1.0 if train_D else 2.0
        noise = self.ts.use_instance_noise
        a = HR + instance_noise(sigma, HR.shape, it, self.niter) if noise else HR
        y_real = nets.discriminator_forward(self.sdD, a, self.ds, training=train_D).squeeze()
        b_in = SR.detach() if train_D else SR
        b = b_in + instance_noise(sigma, HR.shape, it, self.niter) if noise else b_in
        y_fake = nets.discriminator_forward(self.sdD, b, self.ds, training=train_D).squeeze()
        if not train_D:
            y_real = y_real.detach()
        return y_real, y_fake

    # -- G losses: wind_field_GAN_3D.py:342-454 --------------------------------
    def G_loss_terms(self, HR, SR, Z, y_real, y_fake) -> Dict[str, Tensor]:
        ts = self.ts
        bce = F.binary_cross_entropy_with_logits
        if ts.gan_type == "relativistic":
            adv = bce(y_fake - y_real, self.HR_labels)
        elif ts.gan_type == "relativisticavg":
            adv = (bce(y_fake - y_real.mean(), self.HR_labels)
                   + bce(y_real - y_fake.mean(), self.fake_labels)) / 2.0
        else:
            raise NotImplementedError(ts.gan_type)
        pix = torch.zeros(1)
        if ts.pixel_criterion == "l1":
            pix = F.l1_loss(HR, SR)
        elif ts.pixel_criterion == "l2":
            pix = F.mse_loss(HR, SR)
        g_hr = physics.wind_gradient(HR[:, :3], self.x, self.y, Z)
        g_sr = physics.wind_gradient(SR[:, :3], self.x, self.y, Z)
        m_xy, m_z, m_div, m_div2 = physics.gradient_norm_factors(g_hr, g_sr)
        l_xy = F.mse_loss(g_sr[:, :6] / m_xy, g_hr[:, :6] / m_xy)
        l_z = F.mse_loss(g_sr[:, 6:] / m_z, g_hr[:, 6:] / m_z)
        l_div = F.mse_loss((g_hr[:, 0] + g_hr[:, 4] + g_hr[:, 8]) / m_div,
                           (g_sr[:, 0] + g_sr[:, 4] + g_sr[:, 8]) / m_div)
        l_div2 = F.mse_loss((g_hr[:, 0] + g_hr[:, 4]) / m_div2, (g_sr[:, 0] + g_sr[:, 4]) / m_div2)
        terms = {
            "adversarial": adv * ts.w_adv,
            "pix": pix * ts.w_pix,
            "xy_gradient": l_xy * ts.w_gxy,
            "z_gradient": l_z * ts.w_gz,
            "divergence": l_div * ts.w_div,
            "xy_divergence": l_div2 * ts.w_divxy,
            "feature_D": torch.zeros(1),
        }
        phys = [terms[k] for k in ("xy_gradient", "z_gradient", "divergence", "xy_divergence")]
        if any(bool(t.isnan() or t.isinf()) for t in phys):  # :434-445
            total = terms["adversarial"] + terms["pix"] + terms["feature_D"]
        else:
            total = (terms["adversarial"] + terms["pix"] + terms["xy_gradient"] + terms["z_gradient"]
                     + terms["divergence"] + terms["xy_divergence"] + terms["feature_D"])
        terms["total"] = total
        return terms

    def D_loss_value(self, y_real, y_fake) -> Tensor:
        """wind_field_GAN_3D.py:545-562."""
        bce = F.binary_cross_entropy_with_logits
        if self.ts.gan_type == "relativistic":
            return bce(y_real - y_fake, self.HR_labels)
        loss = (bce(y_real - y_fake.mean(), self.HR_labels)
                + bce(y_fake - y_real.mean(), self.fake_labels)) / 2.0
        if torch.all(self.HR_labels == 0.9):
            loss = loss - 0.1985
        return loss

    # -- iterations ----------------------------------------------------------
    def is_G_iteration(self, it: int) -> bool:
        """wind_field_GAN_3D.py:585-587."""
        return (it // self.ts.d_g_train_period) % (self.ts.d_g_train_ratio + 1) == 0

    def G_iteration(self, LR, HR, Z, it: Tensor):
        """update_G(training) :476-490 + calculate_optimize_and_log_G_loss :455-460."""
        SR = nets.generator_forward(self.sdG, LR, Z, self.gs, training=True)
        for p in self.pD:
            p.requires_grad_(False)
        self.opt_G.zero_grad(set_to_none=True)
        y_real, y_fake = self.D_pair(HR, SR, it, train_D=False)
        terms = self.G_loss_terms(HR, SR, Z, y_real, y_fake)
        terms["total"].backward()
        if not bool(terms["total"].isnan() or terms["total"].isinf()):
            self.opt_G.step()
        self.G_losses = {k: v.detach() for k, v in terms.items()}
        return SR.detach()

    def D_iteration(self, LR, HR, Z, it: Tensor):
        """:589-593 then update_D(training) :532-568."""
        with torch.no_grad():
            SR = nets.generator_forward(self.sdG, LR, Z, self.gs, training=False)
        for p in self.pD:
            p.requires_grad_(True)
        self.opt_D.zero_grad(set_to_none=True)
        y_real, y_fake = self.D_pair(HR, SR, it, train_D=True)
        loss = self.D_loss_value(y_real, y_fake)
        loss.backward()
        self.opt_D.step()
        self.D_loss = loss.detach()
        return SR

    def optimize_parameters(self, LR, HR, Z, it: int):
        """compute_losses_and_optimize(training_iteration=True) :570-593."""
        it_t = torch.tensor(it)
        self.make_labels(it_t, HR.shape[0])
        if self.is_G_iteration(it):
            self.G_iteration(LR, HR, Z, it_t)
            return "G"
        self.D_iteration(LR, HR, Z, it_t)
        return "D"

    def update_learning_rate(self):
        for s in self.sched:
            s.step()


def synthetic_batch(B: int, n: int, nz: int, scale: int, seed: int = 2001, in_ch: int = 4):
    """Synthetic HARMONIE-SIMRA-shaped batch, SURVEY.md 8(d).

    HR ~ U(-1,1); LR = strided HR (process_data.py:451,457) + normalised terrain
    channel; Z strictly increasing in z; x = y = 200 m grid.
    """
    g = torch.Generator().manual_seed(seed)
    sn = scale * n
    HR = torch.rand((B, 3, sn, sn, nz), generator=g) * 2 - 1
    col = torch.rand((B, 1, sn, sn, 1), generator=g) * 100.0
    Z = torch.linspace(0.0, 500.0, nz).view(1, 1, 1, 1, nz) + col * torch.linspace(1.0, 0.2, nz).view(1, 1, 1, 1, nz)
    chans = [HR[:, :, ::scale, ::scale, :]]
    if in_ch > 3:
        zc = Z[:, :, ::scale, ::scale, :]
        zc = (zc - Z.min()) / (Z.max() - Z.min())
        # extra input channels (pressure / z / above-ground, wind_field_GAN_3D.py:93-96): the first is the normalised
        # terrain height as before; any further ones are distinct fields (scaled, mirrored) so a channel mix-up shows
        extra = [zc] + [(1.0 - zc) * (0.5 + 0.25 * j) if j % 2 else zc.flip(2) * (0.5 + 0.25 * j) for j in range(1, in_ch - 3)]
        chans.append(torch.cat(extra, dim=1))
    LR = torch.cat(chans, dim=1).contiguous()
    x = torch.arange(sn, dtype=torch.float32) * 200.0
    y = torch.arange(sn, dtype=torch.float32) * 200.0
    return LR, HR.contiguous(), Z.contiguous(), x, y
