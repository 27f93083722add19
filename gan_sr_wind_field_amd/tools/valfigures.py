"""Figures of the validation epoch (reference train.py:236-307, helpers :340-555).

The reference pushes, per validation epoch and for two horizontal slices of one random validation sample - the u
component at level index 3 and a random component at a random level - a *comparison* figure (LR, HR, trilinear
baseline, SR on one colour scale) and an *error* figure (signed error, field, absolute error for SR and for the
trilinear baseline, with the slice's mean absolute errors in the titles) to TensorBoard under
``im/<it>/wind_fields/<title><level>`` and ``im/<it>/Error/<title><level>``.

This module draws the same two figures from the physical-unit (m/s) fields.  ``matplotlib`` is optional, like
``tensorboardX``: without it nothing is drawn; with matplotlib but without a TensorBoard writer the figures are
written as PNG files next to the validation pickles (an addition - the reference only logs to TensorBoard).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np

try:  # optional, never needed by the train step itself
    import matplotlib

    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from matplotlib import cm
except Exception:  # pragma: no cover - exercised only where matplotlib is absent
    plt = None

WIND_COMPONENT = {0: "u", 1: "v", 2: "w"}


def available() -> bool:
    return plt is not None


def _panel(ax, field, title, cmap, lim=None):
    kw = {} if lim is None else {"vmin": lim[0], "vmax": lim[1]}
    ax.pcolormesh(field, cmap=cmap, **kw)
    ax.set_title(title)


def _bar(fig, ax, cmap, lim):
    m = cm.ScalarMappable(cmap=plt.get_cmap(cmap))
    m.set_clim(*lim)
    fig.colorbar(m, ax=ax)


def comparison_figure(level: int, lr, hr, sr, tl):
    """2 x 2 panels - LR, HR / trilinear, SR - of one wind component at model level ``level``, all on the colour
    scale of the HR slice (reference :507-555).  Inputs are (X, Y, Z) arrays in m/s."""
    ref = hr[:, :, level]
    lim = (float(ref.min()), float(ref.max()))
    fig, axes = plt.subplots(2, 2, figsize=(8, 7))
    for ax, (name, f) in zip(axes.ravel(), (("LR", lr), ("HR", hr), ("TL", tl), ("SR", sr))):
        _panel(ax, f[:, :, level], name, "viridis", lim)
    fig.subplots_adjust(hspace=0.3)
    _bar(fig, axes, "viridis", lim)
    return fig


def error_figure(level: int, hr, sr, tl, sr_err: float, tl_err: float):
    """2 x 3 panels: signed error, field, absolute error - first row SR, second row the trilinear baseline
    (reference :383-505).  The SR row's colour scales span both candidates (signed / absolute error) and both fields;
    the baseline row is auto-scaled, as in the reference."""
    h, s, t = hr[:, :, level], sr[:, :, level], tl[:, :, level]
    both = np.concatenate((t, s), axis=0)
    errs = np.concatenate((t - h, s - h), axis=0)
    field_lim = (float(both.min()), float(both.max()))
    err_lim = (float(errs.min()), float(errs.max()))
    abs_lim = (0.0, max(abs(err_lim[0]), abs(err_lim[1])))
    hr_lim = (float(h.min()), float(h.max()))
    fig, axes = plt.subplots(2, 3, figsize=(12, 6), sharex=True, sharey=True)
    _panel(axes[0, 0], s - h, "Error SR-HR (m/s)", "coolwarm", err_lim)
    _panel(axes[0, 1], s, f"SR, avg error: {round(sr_err, 3)} m/s", "viridis", field_lim)
    _panel(axes[0, 2], np.abs(h - s), "SR Absolute Error (m/s)", "jet", abs_lim)
    _panel(axes[1, 0], t - h, "Error TL-HR (m/s)", "coolwarm")
    _panel(axes[1, 1], t, f"TL, avg error: {round(tl_err, 3)} m/s", "viridis")
    _panel(axes[1, 2], np.abs(h - t), "TL Absolute Error (m/s)", "jet")
    for row in (0, 1):
        _bar(fig, axes[row, 0], "coolwarm", err_lim)
        _bar(fig, axes[row, 1], "viridis", hr_lim)
        _bar(fig, axes[row, 2], "jet", abs_lim)
    fig.subplots_adjust(hspace=0.2)
    return fig


def log_validation_figures(imgs: Dict[str, np.ndarray], it: int, tb=None, out_dir: Optional[str] = None,
                           rng: Optional[np.random.Generator] = None) -> list:
    """``imgs`` = {"LR", "HR", "SR", "BC"}: (3, X, Y, Z) fields in m/s of ONE sample (BC = trilinear baseline).
    Draws the reference's two slices, sends the figures to ``tb.add_figure`` when a writer is given, else saves PNGs
    into ``out_dir``.  Returns the tags (nothing is drawn and [] is returned without matplotlib)."""
    if plt is None or (tb is None and out_dir is None):
        return []
    rng = np.random.default_rng() if rng is None else rng
    lr, hr, sr, tl = imgs["LR"], imgs["HR"], imgs["SR"], imgs["BC"]
    nz = hr.shape[-1]
    slices = [(0, min(3, nz - 1)), (int(rng.integers(0, 3)), int(rng.integers(0, nz)))]
    tags = []
    for comp, level in slices:
        sr_err = float(np.abs(hr[comp][:, :, level] - sr[comp][:, :, level]).mean())
        tl_err = float(np.abs(hr[comp][:, :, level] - tl[comp][:, :, level]).mean())
        title = f"{WIND_COMPONENT[comp]}_field_z_index{level}"
        figs = ((f"im/{it}/wind_fields/{title}", comparison_figure(level, lr[comp], hr[comp], sr[comp], tl[comp])),
                (f"im/{it}/Error/{title}", error_figure(level, hr[comp], sr[comp], tl[comp], sr_err, tl_err)))
        for tag, fig in figs:
            if tb is not None:
                tb.add_figure(tag, fig, it)
            else:
                os.makedirs(out_dir, exist_ok=True)
                fig.savefig(os.path.join(out_dir, tag.replace("/", "__") + ".png"), dpi=80)
            plt.close(fig)
            tags.append(tag)
    return tags
