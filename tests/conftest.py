import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(1, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = float(b.norm())
    return float((a - b).norm()) / (den if den > 0 else 1.0)


@pytest.fixture(scope="session")
def hip():
    """The loaded C-ABI library wrapper; GPU tests fail (not skip) if it is missing."""
    from gan_sr_wind_field_amd import _lib

    return _lib.lib()


def reload_wsr_env():
    """the C side caches its WSR_* tuning switches per call site: have them read again after changing os.environ"""
    from gan_sr_wind_field_amd import hip_ops

    hip_ops.reload_env()  # (C side re-reads, host-side plan caches start over)


@pytest.fixture(autouse=True)
def _fresh_wsr_env():
    """a test that changed a WSR_* switch (monkeypatch) must not leak the cached value into the next one"""
    try:
        reload_wsr_env()
    except Exception:  # library not built: the tests that need it fail on their own
        pass
    yield
