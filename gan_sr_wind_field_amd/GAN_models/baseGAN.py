"""BaseGAN: device selection and checkpoint I/O (reference GAN_models/baseGAN.py:19-106).

Checkpoint layout is the reference's: ``G_{it}.pth`` / ``D_{it}.pth`` hold plain
``state_dict``s (same keys, logical (Cout, Cin, kx, ky, kz) fp32 filters) and
``state_{it}.pth`` = ``{"it", "epoch", "schedulers": [...], "optimizers": [...]}``.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from ..tools import loggingclass as lc


def _given(path) -> bool:
    return path is not None and str(path).lower() not in ("null", "none")


class BaseGAN(lc.GlobalLoggingClass):
    G: nn.Module = None
    D: nn.Module = None

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        # ONE source of truth for the device: ``cfg.device`` when a launcher set it (run.py / train.py put
        # every rank on cuda:LOCAL_RANK), else the reference's rule ``cuda:{gpu_id}`` (baseGAN.py:27-33).
        dev = getattr(cfg, "device", None)
        if dev is not None:
            self.device = torch.device(dev)
        else:
            use_gpu = torch.cuda.is_available() and cfg.gpu_id is not None
            self.device = torch.device(f"cuda:{cfg.gpu_id}") if use_gpu else torch.device("cpu")
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.is_train = cfg.is_train
        self.schedulers = []
        self.optimizers = []

    def load_model(self, generator_load_path: str = None, discriminator_load_path: str = None,
                   state_load_path: str = None):
        """Returns ``(epoch, it)`` when a training state was loaded, else ``(None, None)``."""
        if _given(generator_load_path):
            self.G.load_state_dict(torch.load(generator_load_path, map_location="cpu"))
            self.G.eval()
        if _given(discriminator_load_path):
            self.D.load_state_dict(torch.load(discriminator_load_path, map_location="cpu"))
            self.G.eval()
        if _given(state_load_path):
            state = torch.load(state_load_path)
            opts, scheds = state["optimizers"], state["schedulers"]
            assert len(opts) == len(self.optimizers), \
                f"Loaded {len(opts)} optimizers but expected {len(self.optimizers)}"
            assert len(scheds) == len(self.schedulers), \
                f"Loaded {len(scheds)} schedulers but expected {len(self.schedulers)}"
            for mine, theirs in zip(self.optimizers, opts):
                mine.load_state_dict(theirs)
            for mine, theirs in zip(self.schedulers, scheds):
                mine.load_state_dict(theirs)
            return state["epoch"], state["it"]
        return None, None

    def save_model(self, save_basepath: str, epoch: int, it: int, save_G: bool = True, save_D: bool = True,
                   save_state: bool = True):
        folder = self.cfg.env.this_runs_folder  # the argument is ignored, as in the reference (:91)
        if save_G:
            torch.save(self.G.state_dict(), os.path.join(folder, f"G_{it}.pth"))
        if save_D:
            torch.save(self.D.state_dict(), os.path.join(folder, f"D_{it}.pth"))
        if save_state:
            state = {"it": it, "epoch": epoch,
                     "schedulers": [s.state_dict() for s in self.schedulers],
                     "optimizers": [o.state_dict() for o in self.optimizers]}
            torch.save(state, os.path.join(folder, f"state_{it}.pth"))
