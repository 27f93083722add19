"""Training loop for ``wind_field_GAN_3D`` - same entry point, iteration schedule, checkpoint / log /
validation periods and artefacts as the reference's ``train.py:24-337``; the step itself is the HIP path.

    train(cfg, dataset_train, dataset_validation, x, y)

Differences, all additive: ``tensorboardX`` and ``progressbar2`` are optional (scalars still go to the
"train" logger and the validation pickles are still written without them); ``torch.profiler`` wraps the
loop only when ``WSR_TORCH_PROFILER=1`` (the reference always profiles iterations 3-10); and when the
process was started by ``torchrun`` (WORLD_SIZE > 1) the loop runs data-parallel: the sampler shards the
shuffled index list per rank, ``dist.attach`` reduces gradients / BatchNorm statistics / batch-global loss
terms, and only rank 0 writes checkpoints, logs and validation artefacts.
"""
from __future__ import annotations

import contextlib
import logging
import os
import pickle as pkl

import numpy as np
import torch
import torch.nn as nn

from . import dist as wdist
from .GAN_models.wind_field_GAN_3D import wind_field_GAN_3D

try:  # optional, as in many deployments of the reference
    import tensorboardX
except ImportError:  # pragma: no cover - depends on the environment
    tensorboardX = None


class _Bar:
    """progressbar2 when it is installed, otherwise silent (reference iocomponents/displaybar.py)"""

    def __init__(self, max_value, start_epoch, start_it, niter):
        self._bar = None
        try:
            import progressbar
            self._bar = progressbar.ProgressBar(max_value=max_value)
            self._fmt = f"epoch {{}} it {{}}/{niter}"
        except ImportError:
            pass

    def update(self, i, epoch, it):
        if self._bar is not None:
            self._bar.update(i)


def _scalars(d):
    return {k: float(v) for k, v in d.items()}


def train(cfg, dataset_train, dataset_validation, x, y):
    cfg_t = cfg.training
    status_logger = logging.getLogger("status")
    train_logger = logging.getLogger("train")
    distributed = wdist.init_from_env()
    rank = torch.distributed.get_rank() if distributed else 0
    world = torch.distributed.get_world_size() if distributed else 1
    lead = rank == 0
    tb = None
    if lead and tensorboardX is not None and cfg.use_tensorboard_logger:
        tb = tensorboardX.SummaryWriter(log_dir=cfg.env.this_runs_tensorboard_log_folder)

    if not cfg.dataset_train:
        raise ValueError("can't train without a training dataset - adjust the config")
    sampler = None
    if distributed:  # every rank sees a disjoint 1/world of each shuffled epoch
        sampler = torch.utils.data.distributed.DistributedSampler(dataset_train, num_replicas=world, rank=rank,
                                                                  shuffle=True, drop_last=True)
    dataloader_train = torch.utils.data.DataLoader(
        dataset_train, batch_size=cfg.dataset_train.batch_size, shuffle=sampler is None, sampler=sampler,
        num_workers=cfg.dataset_train.num_workers, pin_memory=True, drop_last=distributed)
    status_logger.info("finished creating training dataloader and dataset")
    dataloader_val = None
    if cfg.dataset_val and dataset_validation is not None and len(dataset_validation) > 0:
        dataloader_val = torch.utils.data.DataLoader(
            dataset_validation, batch_size=cfg.dataset_val.batch_size, shuffle=False,
            num_workers=cfg.dataset_val.num_workers, pin_memory=True)
        status_logger.info("finished creating validation dataloader and dataset")
    else:
        status_logger.warning("no validation dataset supplied! consider adjusting the config")

    if cfg.model.lower() != "wind_field_gan_3d":
        raise NotImplementedError(f"only wind_field_GAN_3D is supported - not {cfg.model}")
    gan = wind_field_GAN_3D(cfg)
    status_logger.info(f"Making model wind_field_GAN_3D from config {cfg.name}")
    status_logger.debug(f"GAN:\n{gan}\n")
    for line in gan.get_new_status_logs():
        status_logger.info(line)

    start_epoch, it, loaded_it = 0, 0, 0
    it_per_epoch = max(len(dataloader_train), 1)
    count_train_epochs = 1 + cfg_t.niter // it_per_epoch
    if cfg.load_model_from_save:
        status_logger.info(f"loading model from from saves. G: {cfg.env.generator_load_path}, "
                           f"D: {cfg.env.discriminator_load_path}")
        gan.load_model(generator_load_path=cfg.env.generator_load_path,
                       discriminator_load_path=cfg.env.discriminator_load_path or None, state_load_path=None)
        if cfg_t.resume_training_from_save:
            status_logger.info(f"resuming training from save. state: {cfg.env.state_load_path}")
            loaded_epoch, loaded = gan.load_model(generator_load_path=None, discriminator_load_path=None,
                                                  state_load_path=cfg.env.state_load_path)
            status_logger.info(f"loaded epoch {loaded_epoch}, it {loaded}")
            if loaded:
                start_epoch, it, loaded_it = loaded_epoch, loaded, loaded
    if distributed:
        wdist.attach(gan, bucket_mb=cfg.dist.bucket_mb, sync_bn=cfg.dist.sync_bn)

    bar = _Bar(len(dataloader_train), start_epoch, it, cfg_t.niter) if lead else None
    status_logger.info(f"beginning run from epoch {start_epoch}, it {it}")
    profile = contextlib.nullcontext()
    if os.environ.get("WSR_TORCH_PROFILER") == "1":
        profile = torch.profiler.profile(
            schedule=torch.profiler.schedule(wait=2, warmup=2, active=6, repeat=1),
            on_trace_ready=torch.profiler.tensorboard_trace_handler(cfg.env.this_runs_tensorboard_log_folder),
            with_stack=True, profile_memory=True, record_shapes=True)
    dev = cfg.device
    with profile as profiler:
        for epoch in range(start_epoch, count_train_epochs):
            if sampler is not None:
                sampler.set_epoch(epoch)
            for i, (LR, HR, Z) in enumerate(dataloader_train):
                if it > cfg_t.niter:
                    break
                it += 1
                if bar is not None:
                    bar.update(i, epoch, it)
                LR, HR, Z = (t.to(dev, non_blocking=True) for t in (LR, HR, Z))
                if it == loaded_it + 1:
                    gan.feed_xy_niter(x.to(dev, non_blocking=True), y.to(dev, non_blocking=True),
                                      torch.tensor(cfg_t.niter, device=dev), cfg_t.d_g_train_ratio,
                                      cfg_t.d_g_train_period)
                gan.optimize_parameters(LR, HR, Z, it)
                if profiler is not None:
                    profiler.step()
                if it > 2 * cfg_t.d_g_train_period:
                    gan.update_learning_rate()
                for line in gan.get_new_status_logs():
                    train_logger.info(line)
                if lead and it % cfg_t.save_model_period == 0:
                    status_logger.debug(f"saving model (it {it})")
                    gan.save_model(cfg.env.this_runs_folder, epoch, it)
                if lead and it % cfg_t.log_period == 0:
                    losses = _scalars(gan.get_G_train_loss_dict_ref())
                    train_logger.info(f"it {it} " + " ".join(f"{k}: {v:.6g}" for k, v in losses.items()))
                    if tb is not None:
                        tb.add_scalars("G_loss/train", losses, it)
                if dataloader_val is None or it % cfg_t.val_period != 0:
                    continue
                _validate(cfg, gan, dataloader_val, dataset_train, it, tb, status_logger, lead)
    if tb is not None:
        tb.close()
    return gan


def _validate(cfg, gan, dataloader_val, dataset_train, it, tb, status_logger, lead):
    """Validation epoch (reference :176-336): averages of the G / D losses and metrics over the validation
    set, one random sample stored as physical-unit HR / SR / trilinear / LR fields."""
    status_logger.debug(f"validation epoch (it {it})")
    dev = cfg.device
    G_vals = {k: 0.0 for k in gan.get_G_val_loss_dict_ref()}
    D_vals = {k: 0.0 for k in gan.get_D_loss_dict_ref()}
    M_vals = {k: 0.0 for k in gan.get_metrics_dict_ref()}
    n = len(dataloader_val)
    for LR, HR, Z in dataloader_val:
        LR, HR, Z = (t.to(dev, non_blocking=True) for t in (LR, HR, Z))
        gan.validation(LR, HR, Z, it)
        for acc, src in ((G_vals, gan.get_G_val_loss_dict_ref()), (D_vals, gan.get_D_loss_dict_ref()),
                         (M_vals, gan.get_metrics_dict_ref())):
            for k, v in src.items():
                acc[k] += float(v) / n
    if not lead:
        return
    b = int(torch.randint(LR.shape[0], size=(1,)))
    uvw = float(dataset_train.UVW_MAX)
    LR_i, Z_i = LR[b:b + 1], Z[b:b + 1]
    with torch.no_grad():
        SR_i = (uvw * gan.G(LR_i, Z_i)).squeeze(0)
        TL_i = (uvw * nn.functional.interpolate(LR_i[:, :3], scale_factor=(cfg.scale, cfg.scale, 1), mode="trilinear",
                                                align_corners=True)).squeeze(0)
    imgs = {"HR": (uvw * HR[b]).cpu().numpy(), "SR": SR_i.cpu().numpy(), "BC": TL_i.cpu().numpy(),
            "LR": (uvw * LR_i[0, :3]).cpu().numpy()}
    if cfg.use_tensorboard_logger:
        # the reference's comparison / error figures of two slices (train.py:236-307); PNG files when no TensorBoard
        # writer is installed
        from .tools import valfigures
        valfigures.log_validation_figures(imgs, it, tb=tb, out_dir=os.path.join(cfg.env.this_runs_folder, "images"))
    if tb is not None:
        tb.add_scalars("G_loss/validation", G_vals, it)
        tb.add_scalars("D_loss/", D_vals, it)
        tb.add_scalars("metrics/PSNR", {k: v for k, v in M_vals.items() if "PSNR" in k}, it)
        tb.add_scalars("metrics/pix", {k: v for k, v in M_vals.items() if "pix" in k}, it)
    os.makedirs(os.path.join(cfg.env.this_runs_folder, "images"), exist_ok=True)
    with open(os.path.join(cfg.env.this_runs_folder, "images", f"val_imgs__it_{it}.pkl"), "wb") as f:
        pkl.dump(imgs, f)
    status_logger.debug(f"it: {it} " + " ".join(f"{k}: {v}" for k, v in {**G_vals, **M_vals}.items()))
