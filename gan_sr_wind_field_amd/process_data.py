"""Data side of the wind-field GAN: the physics-gradient operators of the generator loss
(hot path, reference ``process_data.py:273-313``) and the dataset contract that feeds the
train step (reference ``process_data.py:26-270,420-639``, ``download_data.py``).

The reference downloads HARMONIE-SIMRA netCDF files from thredds.met.no and splits them into
one pickle per hour.  There is no network (and no netCDF4) on the MI355X boxes, so the download
is replaced by :func:`write_synthetic_dataset`, which writes smooth synthetic wind fields **in the
reference's own on-disk layout**; everything downstream - file names, normalisation factors,
``CustomizedDataset`` samples ``(LR, HR, Z)``, augmentation, the 80/10/10 split of
``preprosess`` - follows the reference, so real pickles produced by the reference drop in
unchanged:

    ./data/full_dataset_files/static_terrain_x_y.pkl                  [terrain (X,Y), x (X,), y (Y,)]
    ./data/full_dataset_files/<x_0_128_1___y_0_128_1___z_0_10_1>/
        YYYY-MM-DD-HH.pkl      [z, z_above_ground, u, v, w, pressure]   each (X, Y, nz) float64
        max/max_YYYY-MM-DD-HH.pkl   [z_min, z_max, z_above_ground_max, uvw_max, p_min, p_max]
        norm_factors.pkl       [Z_MIN, Z_MAX, Z_ABOVE_GROUND_MAX, UVW_MAX, P_MIN, P_MAX]
    ./data/interpolated_z_data/<same subfolder>/YYYY-MM-DD-HH.pkl     cache of the z-interpolated sample
"""
from __future__ import annotations

import os
import pickle
from datetime import date, datetime, timedelta
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

DATA_ROOT = "./data"


# --------------------------------------------------------------------------- #
# hot path: wind-field derivatives (fp32 torch code on the device)
# --------------------------------------------------------------------------- #
def calculate_div_z(HR_data: torch.Tensor, Z: torch.Tensor) -> torch.Tensor:
    """d/dz on the non-uniform terrain-following levels ``Z`` (B, 1, X, Y, nz).

    Second-order three-point stencil inside, one-sided first differences at the
    lowest / highest level (reference :273-298).
    """
    f = HR_data
    h = Z[..., 1:] - Z[..., :-1]
    below, above = h[..., :-1], h[..., 1:]
    inner = (below ** 2 * f[..., 2:] + (above ** 2 - below ** 2) * f[..., 1:-1] - above ** 2 * f[..., :-2]) / (
        below * above * (below + above))
    bottom = (f[..., 1:2] - f[..., 0:1]) / h[..., 0:1]
    top = (f[..., -1:] - f[..., -2:-1]) / h[..., -1:]
    return torch.cat((bottom, inner, top), dim=-1)


def calculate_gradient_of_wind_field(HR_data: torch.Tensor, x: torch.Tensor, y: torch.Tensor,
                                     Z: torch.Tensor) -> torch.Tensor:
    """(B, 3, X, Y, nz) -> (B, 9, X, Y, nz): d/dx, d/dy (coordinate spacing), d/dz (reference :301-313).
    Device tensors take the fused HIP kernel (forward + adjoint, ``wsr_wind_gradient``); the torch
    expression below is the same arithmetic and serves host-side tensors (tests with stand-in networks)."""
    if HR_data.is_cuda and HR_data.shape[1] == 3 and Z.shape[1] == 1:
        from . import hip_ops
        return hip_ops.wind_gradient(HR_data, x, y, Z)
    grad_x, grad_y = torch.gradient(HR_data, dim=(2, 3), spacing=(x, y))
    return torch.cat((grad_x, grad_y, calculate_div_z(HR_data, Z)), dim=1)


# --------------------------------------------------------------------------- #
# file naming / slicing helpers (reference download_data.py:29-41,258-298,543-564)
# --------------------------------------------------------------------------- #
def filenames_from_start_and_end_dates(start_date: date, end_date: date) -> List[str]:
    """one name per hour of every day in [start_date, end_date]: ``YYYY-MM-DD-HH.pkl``"""
    t0 = datetime(start_date.year, start_date.month, start_date.day)
    hours = ((end_date - start_date).days + 1) * 24
    return [(t0 + timedelta(hours=h)).strftime("%Y-%m-%d-%H") + ".pkl" for h in range(hours)]


def slice_dict_folder_name(x_dict: Dict, y_dict: Dict, z_dict: Dict) -> str:
    part = lambda tag, d: f"{tag}_{d['start']}_{d['max']}_{d['step']}"  # noqa: E731
    return "___".join((part("x", x_dict), part("y", y_dict), part("z", z_dict))) + "/"


def slice_only_dim_dicts(*arrays, x_dict={"start": 4, "max": -4, "step": 1}, y_dict={"start": 4, "max": -3, "step": 1},
                         z_dict={"start": 1, "max": 41, "step": 1}):
    """Slice (X,Y,Z), (X,Y), (T,X,Y,Z) arrays and the x / y coordinate vectors (the first 1-D array is x,
    later ones are y) with ``start:max:step`` dicts."""
    sx = slice(x_dict["start"], x_dict["max"], x_dict["step"])
    sy = slice(y_dict["start"], y_dict["max"], y_dict["step"])
    sz = slice(z_dict["start"], z_dict["max"], z_dict["step"])
    out, seen_1d = [], 0
    for a in arrays:
        if a.ndim == 3:
            out.append(a[sx, sy, sz])
        elif a.ndim == 2:
            out.append(a[sx, sy])
        elif a.ndim == 4:
            out.append(a[:, sx, sy, sz])
        elif a.ndim == 1:
            out.append(a[sx] if seen_1d == 0 else a[sy])
            seen_1d += 1
    return out


# --------------------------------------------------------------------------- #
# z interpolation onto flat above-ground levels (reference download_data.py:301-397)
# --------------------------------------------------------------------------- #
def _interp_columns(new_levels: np.ndarray, old_levels: np.ndarray, values: np.ndarray) -> np.ndarray:
    """``np.interp(new_levels, old_levels[i, j, :], values[i, j, :])`` for every column, vectorised
    (piecewise linear, clamped to the end values outside the old range)."""
    X, Y, nz = values.shape
    old = old_levels.reshape(-1, nz)
    val = values.reshape(-1, nz)
    out = np.empty((old.shape[0], new_levels.size), dtype=values.dtype)
    for k, zq in enumerate(new_levels):  # nz (10..128) iterations over whole (X*Y) vectors
        hi = np.clip((old < zq).sum(axis=1), 1, nz - 1)  # first index with old >= zq
        lo = hi - 1
        rows = np.arange(old.shape[0])
        z0, z1 = old[rows, lo], old[rows, hi]
        t = np.clip((zq - z0) / np.where(z1 > z0, z1 - z0, 1.0), 0.0, 1.0)
        out[:, k] = val[rows, lo] * (1.0 - t) + val[rows, hi] * t
    return out.reshape(X, Y, new_levels.size)


def interpolate_z_axis(x, y, z_above_ground, u, v, w, pressure, terrain):
    """Resample every column from its terrain-following levels to ONE set of above-ground heights
    (linspace between the mean lowest and mean highest level); returns the new z (altitude), the new
    3-D above-ground heights and the interpolated fields."""
    nz = z_above_ground.shape[-1]
    levels = np.linspace(z_above_ground[:, :, 0].mean(), z_above_ground[:, :, -1].mean(), num=nz)
    u, v, w, pressure = (_interp_columns(levels, z_above_ground, f) for f in (u, v, w, pressure))
    new_above = np.broadcast_to(levels, z_above_ground.shape).copy()
    return new_above + terrain[:, :, None], new_above, u, v, w, pressure


def get_interpolated_z_data(filename, x, y, z_above_ground, u, v, w, pressure, terrain):
    """cached :func:`interpolate_z_axis` (one pickle per sample under ./data/interpolated_z_data)"""
    try:
        with open(filename, "rb") as f:
            return tuple(pickle.load(f))
    except (OSError, EOFError, pickle.UnpicklingError):
        out = interpolate_z_axis(x, y, z_above_ground, u, v, w, pressure, terrain)
        os.makedirs(os.path.dirname(filename), exist_ok=True)
        with open(filename, "wb") as f:
            pickle.dump(list(out), f)
        return out


def reverse_interpolate_z_axis(HR_interp: np.ndarray, Z_raw: np.ndarray, Z_interp: np.ndarray) -> torch.Tensor:
    """(B, C, X, Y, nz) fields on interpolated levels back onto the raw levels of every column."""
    out = np.empty_like(HR_interp)
    for b in range(HR_interp.shape[0]):
        for c in range(HR_interp.shape[1]):
            X, Y, nz = HR_interp.shape[2:]
            zi = Z_interp[b, 0].reshape(-1, nz)
            zr = Z_raw[b, 0].reshape(-1, nz)
            vals = HR_interp[b, c].reshape(-1, nz)
            res = np.empty_like(vals)
            for r in range(vals.shape[0]):
                res[r] = np.interp(zr[r], zi[r], vals[r])
            out[b, c] = res.reshape(X, Y, nz)
    return torch.from_numpy(out)


# --------------------------------------------------------------------------- #
# sample -> tensors (reference process_data.py:420-494)
# --------------------------------------------------------------------------- #
def reformat_to_torch(u, v, w, p, z, z_above_ground, Z_MIN, Z_MAX, Z_ABOVE_GROUND_MAX, UVW_MAX, P_MIN, P_MAX,
                      coarseness_factor=4, include_pressure=False, include_z_channel=False,
                      include_above_ground_channel=False, for_plotting=False):
    """HR = (u, v, w) / UVW_MAX;  LR = HR[:, ::s, ::s, :] (+ normalised pressure) (+ normalised height
    channel(s));  Z = raw altitude (1, X, Y, nz).  All float32, layout (C, X, Y, Z)."""
    s = coarseness_factor
    hr = np.stack((u, v, w)) / UVW_MAX
    chans = [hr[:, ::s, ::s, :]]
    if include_pressure:
        pn = ((p - P_MIN) / (P_MAX - P_MIN))[None]
        chans.append(pn[:, ::s, ::s, :])
        if for_plotting:
            hr = np.concatenate((hr, pn))
    if include_z_channel:
        if include_above_ground_channel:
            chans.append((z_above_ground / Z_ABOVE_GROUND_MAX)[None, ::s, ::s, :])
            chans.append(((z - z_above_ground - Z_MIN) / (Z_MAX - Z_MIN - Z_ABOVE_GROUND_MAX))[None, ::s, ::s, :])
        else:
            chans.append(((z - Z_MIN) / (Z_MAX - Z_MIN))[None, ::s, ::s, :])
    lr = np.concatenate(chans)
    return (torch.from_numpy(np.ascontiguousarray(lr)).float(), torch.from_numpy(np.ascontiguousarray(hr)).float(),
            torch.from_numpy(np.ascontiguousarray(z[None])).float())


def _rotate_wind(t: torch.Tensor, quarter_turns: int) -> torch.Tensor:
    """rot90 in the (x, y) plane of a (C, X, Y, Z) sample whose first two channels are the horizontal wind
    components: the vector components rotate with the grid (reference :198-249)."""
    t = torch.rot90(t, quarter_turns, [1, 2]).clone()
    if quarter_turns == 0:
        return t
    a, b = t[0].clone(), t[1].clone()
    if quarter_turns == 1:
        t[0], t[1] = -b, a
    elif quarter_turns == 2:
        t[0], t[1] = -a, -b
    else:
        t[0], t[1] = b, -a
    return t


class CustomizedDataset(torch.utils.data.Dataset):
    """One sample per hourly pickle: ``(LR, HR, Z)`` float32 tensors of layout (C, X, Y, Z)
    (``is_test``: ``(LR, HR, Z, name, HR_raw, Z_raw)``).  Constructor signature as in the reference (:26-51)."""

    def __init__(self, filenames, subfolder_name, Z_MIN, Z_MAX, UVW_MAX, P_MIN, P_MAX, Z_ABOVE_GROUND_MAX, x, y,
                 terrain, include_pressure=False, include_z_channel=False, interpolate_z=False,
                 include_above_ground_channel=False, COARSENESS_FACTOR=4, data_aug_rot=True, data_aug_flip=True,
                 enable_slicing=False, slice_size=64, for_plotting=False, is_test=False):
        self.filenames = list(filenames)
        self.subfolder_name = subfolder_name
        self.Z_MIN, self.Z_MAX, self.Z_ABOVE_GROUND_MAX = Z_MIN, Z_MAX, Z_ABOVE_GROUND_MAX
        self.UVW_MAX, self.P_MIN, self.P_MAX = UVW_MAX, P_MIN, P_MAX
        self.x, self.y, self.terrain = x, y, terrain
        self.include_pressure = include_pressure
        self.include_z_channel = include_z_channel
        self.interpolate_z = interpolate_z
        self.include_above_ground_channel = include_above_ground_channel
        self.coarseness_factor = COARSENESS_FACTOR
        self.data_aug_rot, self.data_aug_flip = data_aug_rot, data_aug_flip
        self.enable_slicing, self.slice_size = enable_slicing, slice_size
        self.for_plotting, self.is_test = for_plotting, is_test
        self.slice_index = 0
        folder = os.path.join(DATA_ROOT, "full_dataset_files", subfolder_name)
        os.makedirs(os.path.join(folder, "max"), exist_ok=True)
        os.makedirs(os.path.join(DATA_ROOT, "interpolated_z_data", subfolder_name), exist_ok=True)
        norm_file = os.path.join(folder, "norm_factors.pkl")
        if not os.path.isfile(norm_file):
            with open(norm_file, "wb") as f:
                pickle.dump([Z_MIN, Z_MAX, Z_ABOVE_GROUND_MAX, UVW_MAX, P_MIN, P_MAX], f)

    def __len__(self) -> int:
        return len(self.filenames)

    def _tensors(self, u, v, w, pressure, z, z_above_ground):
        return reformat_to_torch(u, v, w, pressure, z, z_above_ground, self.Z_MIN, self.Z_MAX, self.Z_ABOVE_GROUND_MAX,
                                 self.UVW_MAX, self.P_MIN, self.P_MAX, coarseness_factor=self.coarseness_factor,
                                 include_pressure=self.include_pressure, include_z_channel=self.include_z_channel,
                                 include_above_ground_channel=self.include_above_ground_channel,
                                 for_plotting=self.for_plotting)

    def __getitem__(self, index):
        name = self.filenames[index]
        with open(os.path.join(DATA_ROOT, "full_dataset_files", self.subfolder_name, name), "rb") as f:
            z, z_above_ground, u, v, w, pressure = pickle.load(f)
        HR_raw = Z_raw = 0
        if self.interpolate_z:
            if self.is_test:  # keep the un-interpolated truth for the evaluation harness
                _, HR_raw, Z_raw = self._tensors(u, v, w, pressure, z, z_above_ground)
            z, z_above_ground, u, v, w, pressure = get_interpolated_z_data(
                os.path.join(DATA_ROOT, "interpolated_z_data", self.subfolder_name, name), self.x, self.y,
                z_above_ground, u, v, w, pressure, self.terrain)
        if self.enable_slicing:
            # U-shaped beta(1/4, 1/4): patches cluster at the domain borders (reference :159-176)
            x0 = round(np.random.beta(0.25, 0.25) * (self.x.size - self.slice_size))
            y0 = round(np.random.beta(0.25, 0.25) * (self.y.size - self.slice_size))
            sx, sy = slice(x0, x0 + self.slice_size), slice(y0, y0 + self.slice_size)
            z, z_above_ground, u, v, w, pressure = (a[sx, sy, :] for a in (z, z_above_ground, u, v, w, pressure))
        LR, HR, Z = self._tensors(u, v, w, pressure, z, z_above_ground)
        if self.data_aug_rot:
            k = int(np.random.randint(0, 4))
            LR, HR, Z = _rotate_wind(LR, k), _rotate_wind(HR, k), torch.rot90(Z, k, [1, 2])
        if self.data_aug_flip:
            for axis, comp in ((1, 0), (2, 1)):  # mirror x -> u changes sign; mirror y -> v changes sign
                if np.random.rand() > 0.5:
                    LR, HR, Z = torch.flip(LR, [axis]), torch.flip(HR, [axis]), torch.flip(Z, [axis])
                    LR[comp], HR[comp] = -LR[comp], -HR[comp]
        if self.is_test:
            return LR, HR, Z, name[:-4], HR_raw, Z_raw
        return LR, HR, Z


# --------------------------------------------------------------------------- #
# synthetic stand-in for the THREDDS download (writes the reference's file layout)
# --------------------------------------------------------------------------- #
def _smooth_field(rng: np.random.Generator, shape: Tuple[int, int], n_modes: int = 6) -> np.ndarray:
    X, Y = shape
    gx, gy = np.meshgrid(np.linspace(0, 1, X), np.linspace(0, 1, Y), indexing="ij")
    f = np.zeros(shape)
    for _ in range(n_modes):
        kx, ky = rng.uniform(0.5, 3.0, 2)
        f += rng.normal() * np.sin(2 * np.pi * (kx * gx + ky * gy) + rng.uniform(0, 2 * np.pi))
    return f / np.sqrt(n_modes)


def synthetic_batch(B: int, n: int, nz: int, scale: int, seed: int = 2001, in_ch: int = 4):
    """One synthetic ``(LR, HR, Z, x, y)`` batch of the dataset contract without touching the disk (bench.py,
    smoke runs): HR ~ U(-1, 1) (real data is wind / UVW_MAX), LR = every ``scale``-th HR column (reference
    process_data.py:451,457) + the normalised terrain-height channel(s) (:477-484), Z strictly increasing
    along z (``calculate_div_z`` divides by the level spacing), x = y = a 200 m grid."""
    g = torch.Generator().manual_seed(seed)
    sn = scale * n
    HR = torch.rand((B, 3, sn, sn, nz), generator=g) * 2 - 1
    ground = torch.rand((B, 1, sn, sn, 1), generator=g) * 100.0
    levels = torch.linspace(0.0, 500.0, nz).view(1, 1, 1, 1, nz)
    Z = levels + ground * torch.linspace(1.0, 0.2, nz).view(1, 1, 1, 1, nz)
    chans = [HR[:, :, ::scale, ::scale, :]]
    if in_ch > 3:
        zc = (Z[:, :, ::scale, ::scale, :] - Z.min()) / (Z.max() - Z.min())
        chans.append(zc.expand(B, in_ch - 3, n, n, nz))
    LR = torch.cat(chans, dim=1).contiguous()
    grid = torch.arange(sn, dtype=torch.float32) * 200.0
    return LR, HR.contiguous(), Z.contiguous(), grid, grid.clone()


def write_synthetic_dataset(start_date: date, end_date: date, x_dict: Dict, y_dict: Dict, z_dict: Dict,
                            seed: int = 2001, overwrite: bool = False) -> str:
    """Write smooth synthetic HARMONIE-SIMRA-like samples for every hour of [start_date, end_date] in the
    reference's layout (module docstring).  Fields: rolling terrain (0-400 m), ~200 m grid spacing,
    terrain-following levels stretched from a few metres to ~550 m above ground, a logarithmic wind profile
    steered by a smooth large-scale flow, terrain-induced vertical velocity, barometric pressure."""
    nx = len(range(x_dict["start"], x_dict["max"], x_dict["step"]))
    ny = len(range(y_dict["start"], y_dict["max"], y_dict["step"]))
    nz = len(range(z_dict["start"], z_dict["max"], z_dict["step"]))
    root = os.path.join(DATA_ROOT, "full_dataset_files")
    sub = slice_dict_folder_name(x_dict, y_dict, z_dict)
    os.makedirs(os.path.join(root, sub, "max"), exist_ok=True)
    rng = np.random.default_rng(seed)
    static = os.path.join(root, "static_terrain_x_y.pkl")
    if overwrite or not os.path.isfile(static):
        terrain = 200.0 + 200.0 * np.tanh(_smooth_field(rng, (nx, ny)))
        with open(static, "wb") as f:
            pickle.dump([terrain, np.arange(nx) * 200.0, np.arange(ny) * 200.0], f)
    with open(static, "rb") as f:
        terrain, x, y = pickle.load(f)
    terrain = terrain[:nx, :ny]
    eta = np.linspace(0.0, 1.0, nz) ** 1.6  # level spacing grows with height
    slope_x, slope_y = np.gradient(terrain, 200.0)
    for name in filenames_from_start_and_end_dates(start_date, end_date):
        path = os.path.join(root, sub, name)
        if os.path.isfile(path) and os.path.isfile(os.path.join(root, sub, "max", "max_" + name)) and not overwrite:
            continue
        top = 520.0 + 30.0 * _smooth_field(rng, (nx, ny))
        z_above = 2.0 + eta[None, None, :] * (top[:, :, None] - 2.0)
        z = z_above + terrain[:, :, None]
        speed = 8.0 + 3.0 * _smooth_field(rng, (nx, ny))
        direction = rng.uniform(0, 2 * np.pi) + 0.4 * _smooth_field(rng, (nx, ny))
        profile = np.log1p(z_above / 0.3) / np.log1p(500.0 / 0.3)  # log law, roughness 0.3 m
        u = (speed * np.cos(direction))[:, :, None] * profile + 0.5 * rng.normal(size=(nx, ny, nz))
        v = (speed * np.sin(direction))[:, :, None] * profile + 0.5 * rng.normal(size=(nx, ny, nz))
        w = (u * slope_x[:, :, None] + v * slope_y[:, :, None]) * np.exp(-z_above / 300.0)
        pressure = 101325.0 * np.exp(-z / 8000.0) + 30.0 * _smooth_field(rng, (nx, ny))[:, :, None]
        with open(path, "wb") as f:
            pickle.dump([z, z_above, u, v, w, pressure], f)
        with open(os.path.join(root, sub, "max", "max_" + name), "wb") as f:
            pickle.dump([float(z.min()), float(z.max()), float(z_above.max()),
                         float(max(u.max(), v.max(), w.max())), float(pressure.min()), float(pressure.max())], f)
    return sub


def download_all_files_and_prepare(start_date: date, end_date: date, x_dict, y_dict, z_dict, terrain,
                                   folder: str = None, train_eval_test_ratio: float = 0.8):
    """File list + normalisation factors (min / max over the TRAINING part of the period only, as in the
    reference :316-417).  Hours whose pickles are missing are generated synthetically - there is no THREDDS
    access on the GPU boxes."""
    folder = folder or os.path.join(DATA_ROOT, "full_dataset_files") + "/"
    names = filenames_from_start_and_end_dates(start_date, end_date)
    sub = slice_dict_folder_name(x_dict, y_dict, z_dict)
    if not all(os.path.isfile(os.path.join(folder, sub, "max", "max_" + n)) for n in names):
        write_synthetic_dataset(start_date, end_date, x_dict, y_dict, z_dict)
    Z_MIN, Z_MAX, UVW_MAX, P_MIN, P_MAX, Z_ABOVE_GROUND_MAX = 10000, 0, 0, 1000000, 0, 0
    for i, n in enumerate(names):
        with open(os.path.join(folder, sub, "max", "max_" + n), "rb") as f:
            z_min, z_max, zag_max, uvw_max, p_min, p_max = pickle.load(f)
        if i < train_eval_test_ratio * len(names):
            Z_MIN, Z_MAX = min(Z_MIN, z_min), max(Z_MAX, z_max)
            UVW_MAX = max(UVW_MAX, uvw_max)
            P_MIN, P_MAX = min(P_MIN, p_min), max(P_MAX, p_max)
            Z_ABOVE_GROUND_MAX = max(Z_ABOVE_GROUND_MAX, zag_max)
    return names, sub, Z_MIN, Z_MAX, Z_ABOVE_GROUND_MAX, UVW_MAX, P_MIN, P_MAX


def preprosess(train_eval_test_ratio=0.8, X_DICT={"start": 0, "max": 128, "step": 1},
               Y_DICT={"start": 0, "max": 128, "step": 1}, Z_DICT={"start": 0, "max": 10, "step": 1},
               start_date=date(2018, 4, 1), end_date=date(2018, 4, 3), include_pressure=True, include_z_channel=False,
               interpolate_z=False, enable_slicing=False, slice_size=64, include_above_ground_channel=False,
               COARSENESS_FACTOR=4, train_aug_rot=False, val_aug_rot=False, train_aug_flip=False, val_aug_flip=False,
               for_plotting=False):
    """-> (dataset_train, dataset_test, dataset_validation, x, y): chronological 80 / 10 / 10 split
    (reference :497-639; the spelling of the name is the reference's)."""
    static = os.path.join(DATA_ROOT, "full_dataset_files", "static_terrain_x_y.pkl")
    if not os.path.isfile(static):
        write_synthetic_dataset(start_date, end_date, X_DICT, Y_DICT, Z_DICT)
    with open(static, "rb") as f:
        terrain, x, y = slice_only_dim_dicts(*pickle.load(f), x_dict=X_DICT, y_dict=Y_DICT)
    names, sub, Z_MIN, Z_MAX, ZAG_MAX, UVW_MAX, P_MIN, P_MAX = download_all_files_and_prepare(
        start_date, end_date, X_DICT, Y_DICT, Z_DICT, terrain, train_eval_test_ratio=train_eval_test_ratio)
    n_train = int(len(names) * train_eval_test_ratio)
    n_test = int(len(names) * (1 - train_eval_test_ratio) / 2)
    common = dict(include_pressure=include_pressure, include_z_channel=include_z_channel, interpolate_z=interpolate_z,
                  include_above_ground_channel=include_above_ground_channel, COARSENESS_FACTOR=COARSENESS_FACTOR,
                  slice_size=slice_size)
    args = (sub, Z_MIN, Z_MAX, UVW_MAX, P_MIN, P_MAX, ZAG_MAX, x, y, terrain)
    dataset_train = CustomizedDataset(names[:n_train], *args, data_aug_rot=train_aug_rot, data_aug_flip=train_aug_flip,
                                      enable_slicing=enable_slicing, for_plotting=for_plotting, **common)
    dataset_test = CustomizedDataset(names[n_train:n_train + n_test], *args, data_aug_rot=False, data_aug_flip=False,
                                     enable_slicing=False, is_test=True, **common)
    dataset_validation = CustomizedDataset(names[n_train + n_test:], *args, data_aug_rot=val_aug_rot,
                                           data_aug_flip=val_aug_flip, enable_slicing=enable_slicing, **common)
    if enable_slicing:  # regular grid: only the spacing matters to the gradient operators
        x, y = x[:slice_size], y[:slice_size]
    return (dataset_train, dataset_test, dataset_validation, torch.from_numpy(np.asarray(x)).float(),
            torch.from_numpy(np.asarray(y)).float())
