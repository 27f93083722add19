"""wind_field_GAN_3D: the ESRGAN-style 3-D wind-field GAN train step on MI355X.

Public surface = the reference's (GAN_models/wind_field_GAN_3D.py:26-814):
``wind_field_GAN_3D(cfg)``, ``feed_xy_niter``, ``optimize_parameters``,
``validation``, ``update_learning_rate``, the ``get_*_dict_ref`` getters with the
same keys, ``save_model`` / ``load_model`` (BaseGAN), ``G`` / ``D`` attributes and
the free functions ``calculate_PSNR``, ``compute_PSNR_for_SR_and_trilinear``,
``get_norm_factors_of_gradients``.  G and D are the HIP-backed networks of
``CNN_models``; the loss algebra (a few dozen reductions over the (B,3,X,Y,Z)
output) is fp32 torch code on the device, the optimizer is ``torch.optim.Adam``
exactly as in the reference.

Data parallelism (one process per GPU, ``torch.distributed``) is an extension:
when a process group is attached (``dist.attach``) the batch-global quantities of
the step - RaGAN logit means, the four gradient normalisers, BatchNorm statistics
- are reduced across ranks so an N-rank step equals the single-GPU step on the
concatenated batch, and G/D gradients are all-reduced in buckets.
"""
from __future__ import annotations

import copy
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.optim.lr_scheduler as lr_scheduler

from ..CNN_models.Discriminator_3D import Discriminator_3D
from ..CNN_models.Generator_3D_Resnet_ESRGAN import Generator_3D
from ..process_data import calculate_gradient_of_wind_field
from ..tools import initialization, trainingtricks
from .baseGAN import BaseGAN

#: D(real) and D(fake) of an iteration as ONE batched pass of the discriminator's feature pyramid (WSR_D_PAIR=0: two)
D_PAIR = __import__("os").environ.get("WSR_D_PAIR", "1") != "0"
# A generator iteration's guards (non-finite loss terms, the normaliser branch) are read from the device AFTER backward has
# been issued for the outcome that is the rule - every term finite - instead of stalling the launch queue in front of it.
# An iteration whose flags then say otherwise is run again from the random-number state it started with (see update_G).
SPECULATE_GUARDS = __import__("os").environ.get("WSR_SPECULATE_GUARDS", "1") != "0"
#: the relativistic-average losses (reference :360-364, :552-556) from one launch each, derivatives included (round 6,
#: wsr_ragan_loss; WSR_FUSED_RAGAN=0: torch's composed ops, ~25 B-element launches per loss and pass)
FUSED_RAGAN = __import__("os").environ.get("WSR_FUSED_RAGAN", "1") != "0"
#: ... and stop speculating for a while when the guards keep firing (a dataset whose physics terms are non-finite on
#: most patches, SR >> HR early in training): every miss costs a whole discarded generator pass, so after this many
#: CONSECUTIVE misses the next SPEC_BACKOFF generator iterations take the careful path (flags read before backward,
#: as the reference does) before speculation is tried again
SPEC_MISS_LIMIT = int(__import__("os").environ.get("WSR_SPEC_MISS_LIMIT", "2"))
SPEC_BACKOFF = int(__import__("os").environ.get("WSR_SPEC_BACKOFF", "64"))


class _GuardsSaidOtherwise(Exception):
    """the speculative generator iteration met a flag that changes the graph: run it again the careful way"""

_G_LOSS_KEYS = ("total", "adversarial", "pix", "xy_gradient", "z_gradient", "divergence", "xy_divergence",
                "feature_D")


def _zeros_dict(keys):
    return {k: torch.zeros(1) for k in keys}


class wind_field_GAN_3D(BaseGAN):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.optimizers, self.schedulers = [], []
        self.train_G_loss_dict = _zeros_dict(_G_LOSS_KEYS)
        self.validation_G_loss_dict = _zeros_dict(_G_LOSS_KEYS)
        self.D_loss_dict = _zeros_dict(("train_loss", "validation_loss"))
        self.hist_dict = {
            "val_grad_G_first_layer": torch.zeros(1), "val_grad_G_last_layer": torch.zeros(1),
            "val_grad_D_first_layer": torch.tensor(-1.0), "val_grad_D_last_layer": torch.tensor(-1.0),
            "val_weight_G_first_layer": torch.zeros(1), "val_weight_G_last_layer": torch.zeros(1),
            "val_weight_D_first_layer": torch.tensor(-1.0), "val_weight_D_last_layer": torch.tensor(-1.0),
            "SR_pix_distribution": torch.zeros(1), "D_pred_HR": torch.zeros(1), "D_pred_SR": torch.zeros(1),
        }
        self.metrics_dict = _zeros_dict(("val_PSNR", "Trilinear_PSNR", "pix_loss_unscaled", "trilinear_pix_loss"))
        self.device_check = ""
        self.batch_size = 1
        dev = getattr(cfg, "device", self.device)
        self.max_diff_squared = torch.tensor(4.0, device=dev)  # HR is in [-1, 1]
        self.epsilon_PSNR = torch.tensor(1e-8, device=dev)
        self.feature_extractor = None
        self.dp = None  # set by dist.attach()

        cfg_G, cfg_gan = cfg.generator, cfg.gan_config
        # The ini's use_mixed_precision flags are parsed but - as in the reference, whose AMP
        # lines are commented out - do not change the arithmetic; [DEFAULT] compute_dtype does.
        bf16 = getattr(cfg, "compute_dtype", "fp32") == "bf16"
        in_ch = (cfg_G.in_num_ch + cfg_gan.include_pressure + cfg_gan.include_z_channel
                 + cfg_gan.include_above_ground_channel)
        self.G = Generator_3D(
            in_ch, cfg_G.out_num_ch, cfg_G.num_features, cfg_G.num_RRDB, upscale=cfg.scale,
            hr_kern_size=cfg_G.hr_kern_size, number_of_RDB_convs=cfg_G.num_RDB_convs,
            RDB_gc=cfg_G.RDB_growth_chan, lff_kern_size=cfg_G.lff_kern_size,
            RDB_residual_scaling=cfg_G.RDB_res_scaling, RRDB_residual_scaling=cfg_G.RRDB_res_scaling,
            act_type=cfg_G.act_type, device=self.device, number_of_z_layers=cfg_gan.number_of_z_layers,
            conv_mode=cfg_gan.conv_mode, use_mixed_precision=bf16,
            terrain_number_of_features=cfg_G.terrain_number_of_features,
            dropout_probability=cfg_G.dropout_probability, max_norm=cfg_G.max_norm,
        ).to(self.device, non_blocking=True)
        initialization.init_weights(self.G, scale=cfg_G.weight_init_scale)
        self.conv_mode = cfg_G.conv_mode
        self.use_D_feature_extractor_cost = cfg_gan.use_D_feature_extractor_cost
        if not cfg.is_train:
            return

        cfg_D, cfg_t = cfg.discriminator, cfg.training
        self.D = Discriminator_3D(
            cfg_D.in_num_ch, cfg_D.num_features, feat_kern_size=cfg_D.feat_kern_size,
            normalization_type=cfg_D.norm_type, act_type=cfg_D.act_type, mode=cfg_D.layer_mode,
            device=self.device, number_of_z_layers=cfg_gan.number_of_z_layers, conv_mode=cfg_gan.conv_mode,
            use_mixed_precision=bf16, enable_slicing=cfg_gan.enable_slicing,
            dropout_probability=cfg_D.dropout_probability,
        ).to(self.device, non_blocking=True)
        initialization.init_weights(self.D, scale=cfg_D.weight_init_scale)

        # the reference's optimizers (:151-162), same state layout and checkpoints; on the GPU the fused
        # multi-tensor implementation of the same update rule is selected (one launch and ~3 ms less host
        # time per step than the foreach default).  It updates parameters without bumping their version
        # counters, so the programs' packed-filter caches are invalidated from a post-step hook.
        fused = {"fused": True} if torch.device(self.device).type == "cuda" else {}
        # (round 4: the same optimizer with its step as ONE launch over a device table of tensor chunks - torch's fused
        # form needs 8 + 1 launches of ~78 workgroups for the generator's 297 tensors; WSR_TABLE_ADAM=0 restores it)
        Adam = torch.optim.Adam
        if fused and os.environ.get("WSR_TABLE_ADAM", "1") != "0":
            from ..tools.table_adam import TableAdam as Adam
        self.optimizer_G = Adam(self.G.parameters(), lr=cfg_t.learning_rate_g,
                                weight_decay=cfg_t.adam_weight_decay_g,
                                betas=(cfg_t.adam_beta1_g, 0.999), **fused)
        self.optimizer_D = Adam(self.D.parameters(), lr=cfg_t.learning_rate_d,
                                weight_decay=cfg_t.adam_weight_decay_d,
                                betas=(cfg_t.adam_beta1_d, 0.999), **fused)
        if fused:
            self.optimizer_G.register_step_post_hook(lambda *_: self.G.program().filters.invalidate())
            self.optimizer_D.register_step_post_hook(lambda *_: self.D.features.program().filters.invalidate())
        self.optimizers += [self.optimizer_G, self.optimizer_D]
        if cfg_t.multistep_lr_steps:
            self.scheduler_G = lr_scheduler.MultiStepLR(self.optimizer_G, cfg_t.multistep_lr_steps,
                                                        gamma=cfg_t.lr_gamma)
            self.scheduler_D = lr_scheduler.MultiStepLR(self.optimizer_D, cfg_t.multistep_lr_steps,
                                                        gamma=cfg_t.lr_gamma)
            self.schedulers += [self.scheduler_G, self.scheduler_D]

        mse = nn.MSELoss
        self.gradient_xy_criterion = mse().to(dev)
        self.gradient_z_criterion = mse().to(dev)
        self.divergence_criterion = mse().to(dev)
        self.xy_divergence_criterion = mse().to(dev)
        self.feature_D_criterion = mse().to(dev)
        crit = cfg_t.pixel_criterion
        if crit is None or crit == "none":
            self.pixel_criterion = None
        elif crit == "l1":
            self.pixel_criterion = nn.L1Loss().to(dev)
        elif crit == "l2":
            self.pixel_criterion = nn.MSELoss().to(dev)
        else:
            raise NotImplementedError(f"Only l1 and l2 (MSE) loss have been implemented for pixel loss, not {crit}")
        if cfg_t.gan_type in ("relativistic", "relativisticavg"):
            self.criterion = nn.BCEWithLogitsLoss().to(dev)
        else:
            raise NotImplementedError(f"Only relativistic and relativisticavg GAN are implemented, not {cfg_t.gan_type}")

    # ------------------------------------------------------------------ inputs
    def feed_xy_niter(self, x: torch.Tensor, y: torch.Tensor, niter: torch.Tensor, d_g_train_ratio: int,
                      d_g_train_period: int):
        self.x, self.y, self.niter = x, y, niter
        # host copy of the iteration budget (ONE read here, none per step): label smoothing and the instance-noise scale
        # are scalar functions of (it, niter) - evaluated on the host in the reference's fp32 tensor arithmetic they cost
        # no launches on the step path (round 6: ~50 of the ~350 ATen micro-launches of a G + D iteration pair)
        self._niter_cpu = niter.detach().to("cpu") if torch.is_tensor(niter) else torch.tensor(niter)
        self.d_g_train_ratio, self.d_g_train_period = d_g_train_ratio, d_g_train_period

    # ------------------------------------------------------------ batch-global ops
    def _mean(self, t: torch.Tensor) -> torch.Tensor:
        """mean over the (global) batch - the RaGAN average logit"""
        return torch.mean(t) if self.dp is None else self.dp.batch_mean(t)

    def _means(self, a: torch.Tensor, b: torch.Tensor):
        """both RaGAN average logits; under data parallelism in ONE collective per pass"""
        return (torch.mean(a), torch.mean(b)) if self.dp is None else self.dp.batch_means(a, b)

    def _ragan(self, u: torch.Tensor, v: torch.Tensor, means_uv) -> torch.Tensor:
        """( BCEWithLogits(u - mean v, HR_labels) + BCEWithLogits(v - mean u, fake_HR_labels) ) / 2 - the generator's
        adversarial term with (u, v) = (D(fake), D(real)) (reference :360-364), the discriminator's loss with (D(real),
        D(fake)) (:552-556).  ``means_uv`` = (mean u, mean v) when the caller already holds the batch-global means.
        CUDA: forward and all derivatives in one launch (``wsr_ragan_loss``; WSR_FUSED_RAGAN=0: the composed ops)."""
        if means_uv is None and self.dp is not None:
            m_u, m_v = self._means(u, v)
        else:
            m_u, m_v = means_uv if means_uv is not None else (None, None)
        if FUSED_RAGAN and u.is_cuda and isinstance(self.criterion, nn.BCEWithLogitsLoss) and u.numel() == v.numel() \
                and u.numel() == self.HR_labels.numel() == self.fake_HR_labels.numel():
            from .. import hip_ops
            return hip_ops.ragan_loss(u, v, self.HR_labels, self.fake_HR_labels, m_u, m_v)
        if m_u is None:
            m_u, m_v = self._means(u, v)
        return (self.criterion(u - m_v, self.HR_labels) + self.criterion(v - m_u, self.fake_HR_labels)) / 2.0

    def _flags(self, flags) -> list:
        """host-side truth of a list of 0-d device flags in one round trip - set when set on ANY rank"""
        f = flags.to(torch.float32) if torch.is_tensor(flags) else torch.stack([v.reshape(()).to(torch.float32) for v in flags])
        if self.dp is not None:
            f = self.dp.global_max(f)
        return [v > 0 for v in f.tolist()]

    def _flags_later(self, flags):
        """the same flags, fetched WITHOUT waiting: the copy to pinned host memory is queued now, the returned callable
        waits for it (by then long finished) and gives the booleans"""
        f = flags.to(torch.float32) if torch.is_tensor(flags) else torch.stack([v.reshape(()).to(torch.float32) for v in flags])
        host = torch.empty(f.shape, dtype=f.dtype, pin_memory=True)
        done = torch.cuda.Event()

        def fetch(v):
            host.copy_(v, non_blocking=True)
            done.record()

        if self.dp is not None:
            # "set on ANY rank": the flags ride in the backward collective of the RaGAN mean logits (a SUM; the first
            # thing the backward pass issues) instead of taking a blocking collective of their own
            self.dp.ride(f, fetch)
        else:
            fetch(f)

        def get():
            if self.dp is not None:
                left = self.dp.take_unridden()
                if left is not None:  # this backward pass held no scalar collective (gan_type "relativistic")
                    fetch(self.dp.global_max(left[0]))
            done.synchronize()
            return [v > 0 for v in host.tolist()]

        return get

    def _noise(self, sigma: float, shape, it):
        if torch.device(self.device).type == "cuda" and getattr(self, "_niter_cpu", None) is not None:
            # scalars on the host (CPU tensors: the reference's own arithmetic), the draw and its scaling on the device
            it_c = it.detach().cpu() if torch.is_tensor(it) else torch.tensor(int(it))
            return trainingtricks.instance_noise(torch.tensor(float(sigma)), shape, it_c, self._niter_cpu, device=self.device)
        return trainingtricks.instance_noise(self._scalar(sigma), shape, it, self.niter, device=self.device)

    def _scalar(self, v, dtype=None) -> torch.Tensor:
        """0-d tensor on the model's device WITHOUT a blocking host-to-device copy: ``torch.tensor(v, device=cuda)``
        copies from pageable memory and waits for the stream to drain - a dozen pipeline bubbles per iteration on
        the reference's label / noise path; ``torch.full`` is a fill kernel."""
        if self.device.type != "cuda":
            return torch.tensor(v, device=self.device) if dtype is None else torch.tensor(v, device=self.device, dtype=dtype)
        if dtype is None:
            dtype = torch.int64 if isinstance(v, int) else torch.float32
        # (read-only constants: one fill per distinct value for the life of the model, not one per use)
        cache = self.__dict__.setdefault("_scalar_cache", {})
        key = (v, dtype)
        if key not in cache:
            if len(cache) > 64:
                cache.clear()
            cache[key] = torch.full((), v, dtype=dtype, device=self.device)
        return cache[key]

    # ------------------------------------------------------------------ D passes
    def D_forward(self, HR: torch.Tensor, fake_HR: torch.Tensor, it: torch.Tensor, train_D: bool):
        """Returns (y_pred, fake_y_pred).  train_D: D.train(), sigma_base 1, fake detached;
        else D.eval(), sigma_base 2, real logits detached (reference :221-304)."""
        if self.device_check == "":
            # the reference builds a debug string here whose last term draws one noise tensor (:228-246)
            self.device_check = str(HR.device) + str(self._noise(2.0, HR.size(), it).device)
        noise_on = self.cfg.training.use_instance_noise
        # D(real) and D(fake) in one batched pass of the feature pyramid (per-call BatchNorm semantics kept, see
        # Discriminator_3D.forward_pair); WSR_D_PAIR=0: two passes
        pair = D_PAIR and hasattr(self.D, "forward_pair") and HR.shape == fake_HR.shape
        if train_D:
            self.D.train()
            real_in = HR + self._noise(1.0, HR.size(), it) if noise_on else HR
            fake = fake_HR.detach()
            fake_in = lambda: fake + self._noise(1.0, HR.size(), it) if noise_on else fake  # noqa: E731 (drawn in call order)
            if pair:
                y_pred, fake_y_pred = (v.squeeze() for v in self.D.forward_pair(real_in, fake_in))
            else:
                y_pred = self.D(real_in).squeeze()
                fake_y_pred = self.D(fake_in()).squeeze()
        else:
            self.D.eval()
            real_in = HR + self._noise(2.0, HR.size(), it) if noise_on else HR
            fake_in = lambda: fake_HR + self._noise(2.0, HR.size(), it) if noise_on else fake_HR  # noqa: E731
            if pair:
                y_pred, fake_y_pred = (v.squeeze() for v in self.D.forward_pair(real_in.detach(), fake_in))
                y_pred = y_pred.detach()
            else:
                y_pred = self.D(real_in).squeeze().detach()
                fake_y_pred = self.D(fake_in()).squeeze()
        return y_pred, fake_y_pred

    # ------------------------------------------------------------------ G losses
    def log_G_losses(self, fake_HR, losses: dict, training_iteration: bool):
        target = self.train_G_loss_dict if training_iteration else self.validation_G_loss_dict
        for k in _G_LOSS_KEYS:
            target[k] = losses[k]
        if not training_iteration:
            self.metrics_dict["pix_loss_unscaled"] = losses["pix"] / self.cfg.training.pixel_loss_weight
            self.hist_dict["SR_pix_distribution"] = fake_HR.detach().cpu().numpy()

    def calculate_optimize_and_log_G_loss(self, HR, fake_HR, Z, y_pred, fake_y_pred, training_iteration: bool):
        t = self.cfg.training
        if t.gan_type not in ("relativistic", "relativisticavg"):
            raise NotImplementedError(f"Only relativistic and relativisticavg GAN are implemented, not {t.gan_type}")
        # Under data parallelism the two batch-global inputs of this loss - the RaGAN mean logits and the eight
        # physics-loss maxima - cross the ranks in ONE collective (dist._MeansMax): the content losses hand their local
        # maxima to `reduce_max`, which returns the global ones and keeps the means for the adversarial term.
        means = []
        reduce_max = None
        if self.dp is not None and t.gan_type == "relativisticavg":
            def reduce_max(m):
                m_a, m_b, g = self.dp.means_and_max(y_pred, fake_y_pred, m)
                means.append((m_a, m_b))
                return g
        pix, l_xy, l_z, l_div, l_div2, sr_branch = self._content_losses(HR, fake_HR, Z, reduce_max=reduce_max)
        if t.gan_type == "relativistic":
            adv = self.criterion(fake_y_pred - y_pred, self.HR_labels)
        else:
            adv = self._ragan(fake_y_pred, y_pred, means[0][::-1] if means else None)

        feat = torch.zeros(1, device=self.device)
        if self.feature_extractor is not None:
            feat = self.feature_D_criterion(self.feature_extractor(HR).detach(), self.feature_extractor(fake_HR))

        # The seven weighted terms as ONE vector product and the guards as ONE finiteness test (was a multiply per term,
        # eight adds and twelve isnan / isinf / or / any launches - the scalar tail of the step is launch-, not work-bound).
        # The reference tests the four physics terms for NaN / Inf (dropping them from the total) and the total
        # itself (skipping the Adam step) in up to 9 host round trips (:434-460); here all flags - plus "a
        # normaliser came from SR" of the fused path - travel in ONE.  Under data parallelism they are OR-ed over
        # the ranks: replicas that took different branches would drift apart for good (parameters are
        # broadcast only once).
        keys = ("adversarial", "feature_D", "pix", "xy_gradient", "z_gradient", "divergence", "xy_divergence")
        wkey = (str(self.device), t.adversarial_loss_weight, t.feature_D_loss_weight, t.pixel_loss_weight, t.gradient_xy_loss_weight,
                t.gradient_z_loss_weight, t.divergence_loss_weight, t.xy_divergence_loss_weight)
        if getattr(self, "_loss_w", (None,))[0] != wkey:
            self._loss_w = (wkey, torch.tensor([float(v) for v in wkey[1:]], dtype=torch.float32).to(self.device))
        L = {}

        def weigh(unweighted):
            # two vectors, two graphs: the total WITHOUT the physics terms (taken when one of them is not finite) must not
            # hang on their graph at all - a zero upstream gradient times a NaN derivative is a NaN in every filter gradient
            wv = self._loss_w[1]
            Lc = torch.stack([v.reshape(()) for v in unweighted[:3]]) * wv[:3]
            Lp = torch.stack([v.reshape(()) for v in unweighted[3:]]) * wv[3:]
            L.update(zip(keys, Lc.unbind() + Lp.unbind()))
            L["feature_D"] = L["feature_D"].reshape(1)  # (the reference's placeholder is torch.zeros(1): totals are (1,))
            return Lc, Lp

        def totals(Lv):
            Lc, Lp = Lv
            core = Lc.sum().reshape(1)
            full = core + Lp.sum()
            bad = ~torch.isfinite(torch.cat([Lp, core, full]))  # [xy, z, div, xydiv | core | full]
            return core, full, torch.cat([bad[:4].any().reshape(1), bad[4:]])

        Lv = weigh([adv, feat, pix, l_xy, l_z, l_div, l_div2])
        core, full, flags = totals(Lv)
        redo = sr_branch if sr_branch is not None else torch.zeros((), dtype=torch.bool, device=core.device)
        if training_iteration and getattr(self, "_speculating", False):
            # backward for "every term finite, normalisers from HR" goes out now; the flags are looked at behind it
            later = self._flags_later(torch.cat([flags, redo.reshape(1)]))
            L["total"] = full
            try:
                full.backward()
            except BaseException:
                # a backward pass that raises (a HIP error an outer loop catches) must not leave the flags waiting for
                # the NEXT scalar collective - the discriminator iteration's, whose length would then differ on this rank
                if self.dp is not None:
                    self.dp.take_unridden()
                raise
            bad, bad_core, bad_full, redo = later()
            if bad or redo:  # another total, or another graph: this pass does not count
                raise _GuardsSaidOtherwise()
            if not bad_full:
                self.optimizer_G.step()
            elif self.dp is not None:
                self.dp.wait()  # the gradient collectives of the skipped step must still complete
            self.log_G_losses(fake_HR, L, training_iteration)
            return full
        bad, bad_core, bad_full, redo = self._flags(torch.cat([flags, redo.reshape(1)]))
        if redo and torch.is_grad_enabled():  # rare: SR a hundred times larger than HR (see _content_losses)
            _, l_xy, l_z, l_div, l_div2, _ = self._content_losses(HR, fake_HR, Z, fused=False)
            core, full, flags = totals(weigh([adv, feat, pix, l_xy, l_z, l_div, l_div2]))
            bad, bad_core, bad_full = self._flags(flags)
        total, total_bad = (core, bad_core) if bad else (full, bad_full)
        L["total"] = total
        if training_iteration:
            total.backward()
            if not total_bad:
                self.optimizer_G.step()
            elif self.dp is not None:
                self.dp.wait()  # the gradient collectives of the skipped step must still complete
        self.log_G_losses(fake_HR, L, training_iteration)
        return total

    def _content_losses(self, HR, fake_HR, Z, fused: bool = True, reduce_max=None):
        """(pix, xy_gradient, z_gradient, divergence, xy_divergence) un-weighted (reference :377-432) and, on the
        fused path, the device flag "a normaliser came from SR" (else None).

        Device tensors with 3 wind components take ONE HIP pass (``wsr_physics_loss_stats``: 6 sums + 8 maxima,
        Jacobians never materialised; every normaliser n is a scalar, so mse(a/n, b/n) = sum (a-b)^2 / (n^2 N))
        and its two-launch backward.  The fused sums are differentiable w.r.t. SR only through the residuals,
        exactly like the reference while n = HR_max; when n = SR_max / 100 (SR a hundred times larger than HR)
        the reference also differentiates the maximum, and the caller re-evaluates the terms with the composed
        ops below (``fused=False``), which keep that path."""
        t = self.cfg.training
        crit = t.pixel_criterion
        if fused and HR.is_cuda and HR.shape[1] == 3 and fake_HR.shape[1] == 3 and Z.shape[1] == 1:
            from .. import hip_ops
            sums, mx = hip_ops.physics_loss_stats(HR, fake_HR, self.x, self.y, Z)
            m = mx.view(2, 4)
            if self.dp is not None:  # (``reduce_max``: the caller's collective, which carries these maxima along)
                m = reduce_max(m) if reduce_max is not None else self.dp.global_max(m)
            n = torch.max(m[0], m[1] / 100)
            nvox = float(HR.shape[0] * HR.shape[2] * HR.shape[3] * HR.shape[4])
            key = (HR.device, nvox)
            if getattr(self, "_mse_den", (None,))[0] != key:  # elements per term: 6, 3, 1, 1 Jacobian channels
                self._mse_den = (key, torch.tensor([6 * nvox, 3 * nvox, nvox, nvox], device=HR.device))
            terms = sums[:4] / (n * n * self._mse_den[1])
            pix = torch.zeros(1, device=self.device)
            if self.pixel_criterion:
                pix = (sums[4] if crit == "l1" else sums[5]) / (3 * nvox)
            return pix, terms[0], terms[1], terms[2], terms[3], (m[1] / 100 > m[0]).any()
        pix = torch.zeros(1, device=self.device)
        if self.pixel_criterion:
            pix = self.pixel_criterion(HR, fake_HR)
        g_hr = calculate_gradient_of_wind_field(HR[:, :3], self.x, self.y, Z)
        g_sr = calculate_gradient_of_wind_field(fake_HR[:, :3], self.x, self.y, Z)
        n_xy, n_z, n_div, n_div2 = get_norm_factors_of_gradients(g_hr, g_sr, self.dp)
        l_xy = self.gradient_xy_criterion(g_sr[:, :6] / n_xy, g_hr[:, :6] / n_xy)
        l_z = self.gradient_z_criterion(g_sr[:, 6:] / n_z, g_hr[:, 6:] / n_z)
        l_div = self.divergence_criterion((g_hr[:, 0] + g_hr[:, 4] + g_hr[:, 8]) / n_div,
                                          (g_sr[:, 0] + g_sr[:, 4] + g_sr[:, 8]) / n_div)
        l_div2 = self.xy_divergence_criterion((g_hr[:, 0] + g_hr[:, 4]) / n_div2,
                                              (g_sr[:, 0] + g_sr[:, 4]) / n_div2)
        return pix, l_xy, l_z, l_div, l_div2, None

    def update_G(self, LR, HR, Z, it, training_iteration: bool):
        if training_iteration and SPECULATE_GUARDS and torch.device(self.device).type == "cuda" \
                and not getattr(self, "_speculating", False) and not getattr(self, "_careful", False):
            # Speculative pass: nothing but gradients is written before the flags are known (D runs in eval mode with
            # frozen parameters, the Adam step comes after them), and the random draws of the pass - Dropout3d masks,
            # instance noise - are taken again from the same generator state if it has to be repeated.
            if getattr(self, "_careful_left", 0) > 0:  # backed off (see SPEC_MISS_LIMIT): the reference's order
                self._careful_left -= 1
                self._careful = True
                try:
                    return self.update_G(LR, HR, Z, it, True)
                finally:
                    self._careful = False
            rng = (torch.get_rng_state(), torch.cuda.get_rng_state(self.device))
            first_call = self.device_check == ""  # the pass about to run makes the reference's one extra noise draw
            self._speculating = True
            try:
                out = self.update_G(LR, HR, Z, it, True)
                self._spec_misses = 0
                return out
            except _GuardsSaidOtherwise:
                if self.dp is not None:
                    self.dp.wait()
                    self.dp.stats.retries += 1  # (the discarded pass's collectives did run: they stay in the ledger)
                torch.set_rng_state(rng[0])
                torch.cuda.set_rng_state(rng[1], self.device)
                if first_call:
                    self.device_check = ""  # ... and the repeated pass makes that draw again, from the restored state
                self.G.zero_grad(set_to_none=True)
                self._spec_misses = getattr(self, "_spec_misses", 0) + 1
                self.spec_retries = getattr(self, "spec_retries", 0) + 1
                if self._spec_misses >= SPEC_MISS_LIMIT > 0:
                    self._careful_left, self._spec_misses = SPEC_BACKOFF, 0
                    if not getattr(self, "_spec_logged", False):
                        self._spec_logged = True
                        self.status_logs.append(
                            f"generator loss guards fired in {SPEC_MISS_LIMIT} consecutive iterations: the next "
                            f"{SPEC_BACKOFF} generator iterations read them before the backward pass (no discarded "
                            f"passes); speculation is retried after that (WSR_SPEC_MISS_LIMIT / WSR_SPEC_BACKOFF)")
                self._speculating, self._careful = False, True
                try:
                    return self.update_G(LR, HR, Z, it, True)
                finally:
                    self._careful = False
            finally:
                self._speculating = False
        if training_iteration:
            self.G.train()
            fake_HR = self.G(LR, Z)
            for p in self.D.parameters():
                p.requires_grad = False
            self.G.zero_grad(set_to_none=True)
            y_pred, fake_y_pred = self.D_forward(HR, fake_HR, it, train_D=False)
            self.calculate_optimize_and_log_G_loss(HR, fake_HR, Z, y_pred, fake_y_pred, True)
        else:
            self.G.eval()
            with torch.no_grad():
                fake_HR = self.G(LR, Z)
                y_pred, fake_y_pred = self.D_forward(HR, fake_HR, it, train_D=False)
                self.calculate_optimize_and_log_G_loss(HR, fake_HR, Z, y_pred, fake_y_pred, False)
        return fake_HR

    # ------------------------------------------------------------------ D update
    def log_D_losses(self, loss_D, y_pred, fake_y_pred, training_epoch):
        if training_epoch:
            self.D_loss_dict["train_loss"] = loss_D
        else:
            self.D_loss_dict["validation_loss"] = loss_D
            self.hist_dict["D_pred_HR"] = torch.sigmoid(y_pred.detach()).cpu().numpy()[np.newaxis]
            self.hist_dict["D_pred_SR"] = torch.sigmoid(fake_y_pred.detach()).cpu().numpy()[np.newaxis]

    def update_D(self, HR: torch.Tensor, fake_HR: torch.Tensor, it, training_epoch: bool):
        if training_epoch:
            for p in self.D.parameters():
                p.requires_grad = True
            self.optimizer_D.zero_grad(set_to_none=True)
            y_pred, fake_y_pred = self.D_forward(HR, fake_HR, it, train_D=True)
        else:
            with torch.no_grad():  # NB D is put in train mode here too, as in the reference (:541-543)
                y_pred, fake_y_pred = self.D_forward(HR, fake_HR, it, train_D=True)
        gan_type = self.cfg.training.gan_type
        if gan_type == "relativistic":
            loss_D = self.criterion(y_pred - fake_y_pred, self.HR_labels)
        elif gan_type == "relativisticavg":
            loss_D = self._ragan(y_pred, fake_y_pred, None)
            # reference (:558-559): ``if torch.all(labels == 0.9): loss_D -= 0.1985`` - the same value without
            # the host round trip of the ``if``
            host = getattr(self, "_labels_all_09", None)
            if host is not None and host[0] is self.HR_labels:  # labels made on the host (make_new_labels): the test is free
                if host[1]:
                    loss_D = loss_D - 0.1985
            else:
                loss_D = loss_D - 0.1985 * torch.all(self.HR_labels == 0.9)
        else:
            raise NotImplementedError(f"Only relativistic and relativisticavg GAN are implemented, not {gan_type}")
        if training_epoch:
            loss_D.backward()
            self.optimizer_D.step()
        self.log_D_losses(loss_D, y_pred, fake_y_pred, training_epoch)

    # ------------------------------------------------------------------ the step
    def compute_losses_and_optimize(self, LR, HR, Z, it, training_iteration: bool = False):
        self.batch_size = HR.size(0)
        it_int = int(it)
        # (CUDA: the iteration number stays a host tensor - its only uses are scalar schedules evaluated on the host)
        it = torch.tensor(it_int) if torch.device(self.device).type == "cuda" else self._scalar(it)
        self.make_new_labels(it)
        t = self.cfg.training
        if self.use_D_feature_extractor_cost and it_int % t.feature_D_update_period == 0:
            self.feature_extractor = copy.deepcopy(self.D.features)
            for p in self.feature_extractor.parameters():
                p.requires_grad = False
        if training_iteration:
            period = it_int // self.d_g_train_period
            if period % (self.d_g_train_ratio + 1) == 0:
                self.update_G(LR, HR, Z, it, True)
            else:
                with torch.no_grad():
                    self.G.eval()
                    fake_HR = self.G(LR, Z)
                self.update_D(HR, fake_HR, it, True)
            return
        fake_HR = self.update_G(LR, HR, Z, it, False)
        self.update_D(HR, fake_HR, it, False)
        self.metrics_dict["val_PSNR"], self.metrics_dict["Trilinear_PSNR"] = compute_PSNR_for_SR_and_trilinear(
            LR, HR, fake_HR, self.max_diff_squared, self.epsilon_PSNR, interpolate=True, device=self.device,
            scale=self.cfg.scale)
        self.metrics_dict["trilinear_pix_loss"] = self.pixel_criterion(HR, _trilinear(LR, self.cfg.scale))

    def optimize_parameters(self, LR, HR, Z, it):
        self.compute_losses_and_optimize(LR, HR, Z, it, training_iteration=True)

    def validation(self, LR, HR, Z, it):
        self.compute_losses_and_optimize(LR, HR, Z, it, training_iteration=False)

    def make_new_labels(self, it):
        """Real / fake label vectors of this iteration (reference :627-678)."""
        t = self.cfg.training
        pred_real, pred_fake = (False, True) if t.flip_labels else (True, False)
        on_host = torch.device(self.device).type == "cuda" and getattr(self, "_niter_cpu", None) is not None
        if on_host:
            # CUDA: both label vectors on the host in the reference's fp32 arithmetic (the normal draw is a CPU draw there
            # too, trainingtricks.py:37-39), ONE pinned upload for the pair - was ~25 fill / add / clamp launches
            it_c = it.detach().cpu() if torch.is_tensor(it) else torch.tensor(int(it))
            niter, dev, mk = self._niter_cpu, torch.device("cpu"), torch.tensor
        else:
            it_c, niter, dev, mk = it, self.niter, self.device, self._scalar
        real = mk(1.0)
        fake = mk(0.0)
        if t.use_one_sided_label_smoothing and t.flip_labels:
            fake = mk(0.1) - 0.1 * it_c / niter
        elif t.use_one_sided_label_smoothing:
            real = mk(0.9) + 0.1 * it_c / niter
        extra = {} if t.use_noisy_labels else {"noise_stddev": 0.0}
        a = trainingtricks.noisy_labels(pred_real, self.batch_size, true_label_val=real, false_label_val=fake, device=dev, **extra)
        b = trainingtricks.noisy_labels(pred_fake, self.batch_size, true_label_val=real, false_label_val=fake, device=dev, **extra)
        if on_host:
            up = torch.stack([a.reshape(-1), b.reshape(-1)]).pin_memory().to(self.device, non_blocking=True)
            self.HR_labels, self.fake_HR_labels = up[0].squeeze(), up[1].squeeze()
            self._labels_all_09 = (self.HR_labels, bool(torch.all(a == 0.9)))  # (update_D's reference test, known here)
        else:
            self.HR_labels, self.fake_HR_labels = a.squeeze(), b.squeeze()
            self._labels_all_09 = None

    # ------------------------------------------------------------------ getters
    def get_G_train_loss_dict_ref(self):
        return self.train_G_loss_dict

    def get_G_val_loss_dict_ref(self):
        return self.validation_G_loss_dict

    def get_D_loss_dict_ref(self):
        return self.D_loss_dict

    def get_hist_dict_ref(self):
        return self.hist_dict

    def get_metrics_dict_ref(self):
        return self.metrics_dict

    def update_learning_rate(self):
        for s in self.schedulers:
            s.step()

    def count_params(self):
        return (sum(p.numel() for p in self.G.parameters()), sum(p.numel() for p in self.D.parameters()))

    def count_trainable_params(self):
        return (sum(p.numel() for p in self.G.parameters() if p.requires_grad),
                sum(p.numel() for p in self.D.parameters() if p.requires_grad))

    def __str__(self):
        g, d = self.count_params()
        gt, dt = self.count_trainable_params()
        return (f"*---------------*\nGenerator:\n{g} params, {gt} trainable\n\n{self.G}\n\n"
                f"*---------------*\nDiscriminator:\n{d} params, {dt} trainable\n\n{self.D}\n")


# ---------------------------------------------------------------------- metrics
def _trilinear(LR, scale):
    return nn.functional.interpolate(LR[:, :3], scale_factor=(scale, scale, 1), mode="trilinear",
                                     align_corners=True)


def calculate_PSNR(HR: torch.Tensor, fake_HR: torch.Tensor, max_diff_squared=torch.tensor(4.0),
                   epsilon_PSNR=torch.tensor(1e-8), device=torch.device("cpu")):
    """10 log10(max^2 / (MSE + eps)), MSE averaged over B*X*Y*Z - channels are summed (reference :730-742)."""
    w, h, l = HR.shape[2], HR.shape[3], HR.shape[4]
    mse = torch.sum((HR - fake_HR) ** 2) / (w * h * l * HR.shape[0])
    return torch.tensor(10, device=device) * math.log10(max_diff_squared / (mse + epsilon_PSNR))


def compute_PSNR_for_SR_and_trilinear(LR, HR, fake_HR, max_diff_squared, epsilon_PSNR, interpolate: bool = False,
                                      device=torch.device("cpu"), scale: int = 4):
    val = calculate_PSNR(HR, fake_HR, max_diff_squared, epsilon_PSNR, device=device)
    if not interpolate:
        return val
    return val, calculate_PSNR(HR, _trilinear(LR, scale), max_diff_squared, epsilon_PSNR, device=device)


def get_norm_factors_of_gradients(HR_wind_gradient: torch.Tensor, SR_wind_gradient: torch.Tensor, dp=None):
    """[xy-gradient, z-gradient, divergence, xy-divergence] normalisers = max(HR_max, SR_max / 100).

    Batch-global maxima; the z-gradient maximum is taken WITHOUT abs, like the
    reference (:780-781).  Under data parallelism the 8 maxima are max-reduced
    across ranks in one collective that keeps them differentiable (``dist._GlobalMax``).
    """
    def stats(g):
        div3 = g[:, 0] + g[:, 4] + g[:, 8]
        div2 = g[:, 0] + g[:, 4]
        return torch.stack([g[:, :6].abs().max(), g[:, 6:].max(), div3.abs().max(), div2.abs().max()])

    m = torch.stack([stats(HR_wind_gradient), stats(SR_wind_gradient)])
    if dp is not None:  # stays in the graph like the single-GPU maxima (gradient goes to the owning rank)
        m = dp.global_max(m)
    out = torch.max(m[0], m[1] / 100)
    return [out[0], out[1], out[2], out[3]]
