"""Evaluation of a trained generator (``run.py --test``): per-field PSNR / error metrics of the
super-resolved wind field against HR and against the trilinear baseline, written to the same CSV files
and pickles as the reference's ``test.py:22-374``; the generator forward is the HIP path.

Artefacts (relative to the working directory / the run folder, as in the reference):
    ./test_output/<cfg.name>____metrics.csv                       one row per field
    ./test_output/averages.csv                                     one appended row per run
    ./test_output/<cfg.name>____metrics_reverse_interpolate.csv    (+ averages_reverse_interpolate.csv)
        when ``interpolate_z`` and ``reverse_interpolate``: metrics on the raw terrain-following levels
    <run folder>/fields/test_fields_<name>.pkl                     HR / SR / TL / LR / Z every log_period-th batch
"""
from __future__ import annotations

import logging
import os
import pickle as pkl

import torch
import torch.nn as nn

from .GAN_models.wind_field_GAN_3D import calculate_PSNR, wind_field_GAN_3D
from .process_data import reverse_interpolate_z_axis

METRIC_NAMES = ("PSNR", "PSNR_trilinear", "relative_error", "pix", "trilinear_pix", "relative_error_trilinear",
                "average_wind_speed", "old_pix", "old_pix_trilinear")


def field_metrics(HR: torch.Tensor, SR: torch.Tensor, trilinear: torch.Tensor, UVW_MAX: float) -> dict:
    """Metrics of one (1, 3, X, Y, Z) field in normalised units (reference ``write_metrics`` :334-374):
    PSNR, mean length of the error vector in m/s ("pix"), the same relative to the mean wind speed, and the
    component-wise L1 error ("old pix"), each for the network and for the trilinear baseline."""
    def vec_len(t):
        return torch.sqrt(t[:, 0] ** 2 + t[:, 1] ** 2 + t[:, 2] ** 2).mean()

    err, err_tl, speed = vec_len(HR - SR), vec_len(HR - trilinear), vec_len(HR)
    l1 = nn.L1Loss()
    return {
        "PSNR": float(calculate_PSNR(HR, SR)), "PSNR_trilinear": float(calculate_PSNR(HR, trilinear)),
        "relative_error": float(err / speed), "pix": float(err * UVW_MAX), "trilinear_pix": float(err_tl * UVW_MAX),
        "relative_error_trilinear": float(err_tl / speed), "average_wind_speed": float(speed * UVW_MAX),
        "old_pix": float(l1(HR, SR) * UVW_MAX), "old_pix_trilinear": float(l1(HR, trilinear) * UVW_MAX),
    }


def write_metrics(HR, SR, trilinear, field_name, dest_file, UVW_MAX):
    m = field_metrics(HR, SR, trilinear, UVW_MAX)
    dest_file.write(f"{field_name}," + ",".join(str(m[k]) for k in METRIC_NAMES) + "\n")
    return tuple(m[k] for k in METRIC_NAMES)


def write_fields(LR, HR, SR, interpolated_LR, Z, folder_path, field_name, rawHR=None, Z_raw=None, SR_orig=None):
    fields = {"HR": HR, "SR": SR, "TL": interpolated_LR, "LR": LR, "Z": Z}
    if rawHR is not None and torch.is_tensor(rawHR) and rawHR.numel() > 0:
        fields.update({"HR_orig": rawHR, "Z_orig": Z_raw, "SR_orig": SR_orig})
    fields = {k: (v.squeeze().cpu().numpy() if torch.is_tensor(v) else v) for k, v in fields.items() if v is not None}
    os.makedirs(os.path.join(folder_path, "fields"), exist_ok=True)
    with open(os.path.join(folder_path, "fields", f"test_fields_{field_name}.pkl"), "wb") as f:
        pkl.dump(fields, f)


def _header(path: str, line: str) -> None:
    if not os.path.exists(path):
        with open(path, "w") as f:
            f.write(line + "\n")


def test(cfg, dataset_test, reverse_interpolate: bool = False):
    log = logging.getLogger("status")
    if cfg.dataset_test is None:
        raise ValueError("Test dataset not supplied")
    loader = torch.utils.data.DataLoader(dataset_test, batch_size=1, shuffle=False,
                                         num_workers=min(8, os.cpu_count() or 1), pin_memory=True)
    if cfg.model.lower() != "wind_field_gan_3d":
        raise NotImplementedError(f"only wind_field_GAN_3D is supported - not {cfg.model}")
    gan = wind_field_GAN_3D(cfg)
    log.info(f"loading model from from saves. G: {cfg.env.generator_load_path}")
    gan.load_model(generator_load_path=cfg.env.generator_load_path, discriminator_load_path=None, state_load_path=None)
    gan.G.eval()
    if not reverse_interpolate:
        cfg.gan_config.interpolate_z = False
    rev = bool(cfg.gan_config.interpolate_z)
    uvw = float(dataset_test.UVW_MAX)
    os.makedirs("./test_output", exist_ok=True)
    os.makedirs(os.path.join(cfg.env.this_runs_folder, "fields"), exist_ok=True)
    cols = "field," + ",".join(METRIC_NAMES)
    _header("./test_output/averages.csv", "Name," + ",".join("Average " + k for k in METRIC_NAMES))
    metrics_path = os.path.join("./test_output", cfg.name + "____metrics.csv")
    rev_path = os.path.join("./test_output", cfg.name + "____metrics_reverse_interpolate.csv")
    if rev:
        _header("./test_output/averages_reverse_interpolate.csv",
                "Name," + ",".join("Average " + k for k in METRIC_NAMES))
    n = max(len(dataset_test), 1)
    avg = {k: 0.0 for k in METRIC_NAMES}
    avg_rev = {k: 0.0 for k in METRIC_NAMES}
    dev = cfg.device
    log.info("beginning test")
    with open(metrics_path, "w") as out, (open(rev_path, "w") if rev else open(os.devnull, "w")) as out_rev:
        out.write(cols + "\n")
        out_rev.write(cols + "\n")
        for j, (LR, HR, Z, names, HR_raw, Z_raw) in enumerate(loader):
            TL = nn.functional.interpolate(LR[:, :3], scale_factor=(cfg.scale, cfg.scale, 1), mode="trilinear",
                                           align_corners=True)
            for i in range(LR.shape[0]):
                with torch.no_grad():
                    SR_i = gan.G(LR[i:i + 1].to(dev, non_blocking=True), Z[i:i + 1].to(dev, non_blocking=True)).cpu()
                HR_i, TL_i = HR[i:i + 1, :3], TL[i:i + 1]
                if rev:  # back onto the raw terrain-following levels of every column
                    SR_r = reverse_interpolate_z_axis(SR_i.numpy(), Z_raw[i:i + 1].numpy(), Z[i:i + 1].numpy())
                    TL_r = reverse_interpolate_z_axis(TL_i.numpy(), Z_raw[i:i + 1].numpy(), Z[i:i + 1].numpy())
                    vals = write_metrics(HR_raw[i:i + 1, :3], SR_r, TL_r, names[i], out_rev, uvw)
                    for k, v in zip(METRIC_NAMES, vals):
                        avg_rev[k] += v / n
                vals = write_metrics(HR_i, SR_i, TL_i, names[i], out, uvw)
                for k, v in zip(METRIC_NAMES, vals):
                    avg[k] += v / n
                if j % cfg.training.log_period == 0:
                    write_fields(LR[i], HR[i], SR_i[0], TL[i], Z[i], cfg.env.this_runs_folder, names[i],
                                 HR_raw[i] if rev else None, Z_raw[i] if rev else None, None)
    with open("./test_output/averages.csv", "a") as f:
        f.write(cfg.name + "," + ",".join(str(avg[k]) for k in METRIC_NAMES) + "\n")
    for k in METRIC_NAMES:
        log.info(f"Average {k}: {avg[k]}")
    if rev:
        with open("./test_output/averages_reverse_interpolate.csv", "a") as f:
            f.write(cfg.name + "," + ",".join(str(avg_rev[k]) for k in METRIC_NAMES) + "\n")
    return avg
