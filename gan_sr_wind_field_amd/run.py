"""Entry point: ``python run.py < --train | --test | --use > [--cfg path/to/config.ini]`` - the flags,
run-folder layout, log files, seeding and ``config.ini`` snapshot of the reference's ``run.py:29-319``.

Not available here (no network, no ray/optuna): ``--download`` and ``--param_search`` raise
``NotImplementedError``; missing data files are generated synthetically in the reference's format
(``process_data.write_synthetic_dataset``).  Multi-GPU: ``python -m torch.distributed.run --nproc-per-node N
run.py --train ...`` trains data-parallel (one process per GPU, RCCL).
"""
from __future__ import annotations

import argparse
import logging
import os
import random
from datetime import date

import numpy as np
import torch

from .config.config import Config
from .process_data import preprosess

_PKG = os.path.dirname(os.path.abspath(__file__))


def argv_to_cfg(argv=None) -> Config:
    ap = argparse.ArgumentParser(description="Set config, and set if we're doing training or testing.")
    ap.add_argument("--cfg", type=str, default=os.path.join(_PKG, "config", "wind_field_GAN_3D_config_local.ini"),
                    help="path to config ini file (defaults to config/wind_field_GAN_3D_config_local.ini)")
    for flag, text in (("--train", "run training with supplied config"), ("--test", "run tests with supplied config"),
                       ("--param_search", "hyper-parameter search (needs ray + optuna: not available)"),
                       ("--use", "use on LR images"), ("--download", "only download data (no network: not available)"),
                       ("--loglevel", "unused (kept for command-line compatibility)")):
        ap.add_argument(flag, default=False, action="store_true", help=text)
    ap.add_argument("--slurm_array_id", type=int, default=1, help="ID for slurm job")
    args = ap.parse_args(argv)
    cfg_path = os.path.join(_PKG, "config", "config_use.ini") if args.use else args.cfg
    cfg = Config(cfg_path)
    cfg.is_test, cfg.is_use, cfg.is_train = args.test, args.use, args.train
    cfg.is_download, cfg.is_param_search = args.download, args.param_search
    cfg.slurm_array_id = args.slurm_array_id
    return cfg


def makedirs(path: str) -> None:
    os.makedirs(path, exist_ok=True)


def setup_seed(seed: int) -> None:
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


def safe_setup_env_and_cfg(cfg: Config) -> bool:
    env = cfg.env
    env.log_folder = env.root_path + env.log_subpath
    env.tensorboard_log_folder = env.root_path + env.tensorboard_subpath
    env.status_log_file = env.log_folder + "/" + cfg.name + ".log"
    env.this_runs_folder = env.root_path + env.runs_subpath + "/" + cfg.name
    env.this_runs_tensorboard_log_folder = env.tensorboard_log_folder + "/" + cfg.name
    env.train_log_file = env.this_runs_folder + "/" + cfg.name + ".train"
    for p in (env.log_folder, env.tensorboard_log_folder, "./data/downloaded_raw_bessaker_data",
              "./data/full_dataset_files", "./data/interpolated_z_data", env.this_runs_folder + "/images",
              env.this_runs_tensorboard_log_folder):
        makedirs(p)
    setup_seed(env.fixed_seed)
    return True


def setup_torch(cfg: Config) -> None:
    """``cuda:{gpu_id}`` when a GPU is present and ``gpu_id`` is set (under torchrun: this rank's GPU),
    else cpu - where the HIP networks refuse to run."""
    local = os.environ.get("LOCAL_RANK")
    gpu = int(local) if local is not None else cfg.gpu_id
    cfg.device = torch.device(f"cuda:{gpu}") if torch.cuda.is_available() and gpu is not None else torch.device("cpu")
    if cfg.device.type == "cuda":
        cfg.gpu_id = gpu  # keep the ini-level field in step with the device every rank really uses
        torch.cuda.set_device(cfg.device)


def save_config(cfg: Config, folder: str) -> None:
    if cfg.env.discriminator_load_path is None:
        n = str(cfg.training.niter)
        cfg.env.discriminator_load_path = folder + "/D_" + n + ".pth"
        cfg.env.generator_load_path = folder + "/G_" + n + ".pth"
        cfg.env.state_load_path = folder + "/state_" + n + ".pth"
    with open(folder + "/config.ini", "w") as ini:
        ini.write(cfg.asINI())


def setup_logger(cfg: Config) -> None:
    root = logging.getLogger("status")
    root.setLevel(logging.DEBUG)
    fmt = logging.Formatter("%(asctime)s - %(levelname)s - %(filename)s: %(message)s")
    if cfg.is_train:
        h = logging.FileHandler(cfg.env.status_log_file, mode="a")
        h.setFormatter(fmt)
        h.setLevel(logging.DEBUG)
        root.addHandler(h)
        train_logger = logging.getLogger("train")
        train_logger.setLevel(logging.INFO)
        th = logging.FileHandler(cfg.env.train_log_file, mode="a")
        th.setFormatter(logging.Formatter("%(message)s"))
        train_logger.addHandler(th)
        train_logger.info("Initialized train logger")
    if cfg.also_log_to_terminal:
        t = logging.StreamHandler()
        t.setFormatter(fmt)
        t.setLevel(logging.INFO)
        root.addHandler(t)
    root.info("Initialized status logger")


def prepare_data(cfg: Config):
    g = cfg.gan_config
    return preprosess(
        Z_DICT={"start": 0, "max": g.number_of_z_layers, "step": 1}, start_date=date(*g.start_date),
        end_date=date(*g.end_date), include_pressure=g.include_pressure, include_z_channel=g.include_z_channel,
        interpolate_z=g.interpolate_z, enable_slicing=g.enable_slicing, slice_size=g.slice_size,
        include_above_ground_channel=g.include_above_ground_channel, train_aug_rot=cfg.dataset_train.data_aug_rot,
        train_aug_flip=cfg.dataset_train.data_aug_flip, val_aug_rot=cfg.dataset_val.data_aug_rot,
        val_aug_flip=cfg.dataset_val.data_aug_flip, train_eval_test_ratio=cfg.training.train_eval_test_ratio,
        COARSENESS_FACTOR=cfg.scale)


def main(argv=None) -> None:
    cfg = argv_to_cfg(argv)
    if not (cfg.is_test or cfg.is_train or cfg.is_use or cfg.is_download or cfg.is_param_search):
        print("pass either --test, --download, --use or --train as args, and optionally --cfg path/to/config.ini "
              "if config/wind_field_GAN_3D_config_local.ini isn't what you're planning on using.")
        return
    if cfg.is_download:
        raise NotImplementedError("--download needs thredds.met.no and netCDF4; this build generates synthetic "
                                  "HARMONIE-SIMRA-format samples instead (process_data.write_synthetic_dataset)")
    if cfg.is_param_search:
        raise NotImplementedError("--param_search needs ray.tune and optuna, which are outside this build")
    if not safe_setup_env_and_cfg(cfg):
        print("Aborting")
        return
    setup_torch(cfg)
    save_config(cfg, cfg.env.this_runs_folder)
    setup_logger(cfg)
    log = logging.getLogger("status")
    log.info(f"run.py: initialized with config:\n\n{cfg}")
    log.info(f"run.py: running with device: {cfg.device}")
    dataset_train, dataset_test, dataset_validation, x, y = prepare_data(cfg)
    log.info("run.py: data prepared")
    from .test import test
    from .train import train

    if cfg.is_train:
        log.info("run.py: starting training" + ("" if not cfg.is_test else " before testing"))
        train(cfg, dataset_train, dataset_validation, x, y)
        log.info("run.py: finished training")
        cfg.is_train = False
    if cfg.is_test or cfg.is_use:
        log.info("run.py: starting testing")
        test(cfg, dataset_test)
        log.info("run.py: finished testing")
    log.info(f"run.py: log file location: {cfg.env.status_log_file}  run file location: {cfg.env.train_log_file}")


if __name__ == "__main__":
    main()
