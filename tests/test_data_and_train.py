"""CPU tests of the rows next to the hot path: dataset contract + synthetic
HARMONIE-SIMRA writer, the train loop / CLI plumbing and the evaluation harness (networks answered by
the CPU oracle stand-ins of tests/oracle_nets.py)."""
import os
import pickle
from datetime import date

import numpy as np
import pytest
import torch

from conftest import REPO

LOCAL_INI = os.path.join(REPO, "gan_sr_wind_field_amd", "config", "wind_field_GAN_3D_config_local.ini")
XD = {"start": 0, "max": 128, "step": 1}
ZD = {"start": 0, "max": 10, "step": 1}


@pytest.fixture()
def data_root(tmp_path, monkeypatch):
    from gan_sr_wind_field_amd import process_data as pd

    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(pd, "DATA_ROOT", str(tmp_path / "data"))
    return tmp_path / "data"


def test_synthetic_dataset_layout_and_validity(data_root):
    from gan_sr_wind_field_amd import process_data as pd

    sub = pd.write_synthetic_dataset(date(2018, 3, 1), date(2018, 3, 1), XD, XD, ZD, seed=7)
    assert sub == "x_0_128_1___y_0_128_1___z_0_10_1/"
    names = pd.filenames_from_start_and_end_dates(date(2018, 3, 1), date(2018, 3, 1))
    assert names[0] == "2018-03-01-00.pkl" and names[-1] == "2018-03-01-23.pkl" and len(names) == 24
    terrain, x, y = pickle.load(open(data_root / "full_dataset_files" / "static_terrain_x_y.pkl", "rb"))
    assert terrain.shape == (128, 128) and x.shape == (128,) and np.allclose(np.diff(x), 200.0)
    z, zag, u, v, w, p = pickle.load(open(data_root / "full_dataset_files" / sub / names[5], "rb"))
    for a in (z, zag, u, v, w, p):
        assert a.shape == (128, 128, 10) and a.dtype == np.float64 and np.isfinite(a).all()
    assert np.allclose(z - zag, terrain[:, :, None])            # altitude = terrain + height above ground
    assert (np.diff(z, axis=-1) > 0).all()                      # strictly increasing levels (d/dz divides by dZ)
    assert max(abs(u).max(), abs(v).max(), abs(w).max()) < 100 and p.max() < 200000  # reference validity filter
    mx = pickle.load(open(data_root / "full_dataset_files" / sub / "max" / ("max_" + names[5]), "rb"))
    assert len(mx) == 6 and mx[0] == z.min() and mx[1] == z.max() and mx[3] == max(u.max(), v.max(), w.max())


def test_dataset_contract_vs_reference_fixture(golden, data_root):
    """process_data against fixtures the REFERENCE's own process_data / download_data produced on the same files
    (tests/golden/dataset_contract.npz, make_golden.py gen_data): split and normalisation factors of ``preprosess``,
    ``reformat_to_torch`` channel layouts, z-interpolation (+ its inverse), beta slice sampling and the rot90 / flip
    augmentation with the u, v sign rules under the same seeded numpy RNG stream (reference :159-262, :420-494)."""
    from gan_sr_wind_field_amd import process_data as pd

    g = golden("dataset_contract.npz")
    XS, ZS = {"start": 0, "max": 32, "step": 1}, {"start": 0, "max": 6, "step": 1}
    d0, d1 = date(2018, 3, 1), date(2018, 3, 2)
    pd.write_synthetic_dataset(d0, d1, XS, XS, ZS, seed=2001)
    T = torch.from_numpy
    for tag, kw in (("slice_aug", dict(include_pressure=False, include_z_channel=True, interpolate_z=False,
                                        enable_slicing=True, slice_size=16, train_aug_rot=True, train_aug_flip=True)),
                    ("interp", dict(include_pressure=True, include_z_channel=True, interpolate_z=True,
                                    include_above_ground_channel=True, enable_slicing=False))):
        tr, te, va, x, y = pd.preprosess(X_DICT=XS, Y_DICT=XS, Z_DICT=ZS, start_date=d0, end_date=d1,
                                         COARSENESS_FACTOR=4, **kw)
        assert [len(tr), len(te), len(va)] == list(g[f"{tag}.n"])
        assert [tr.filenames[0], te.filenames[0], va.filenames[0]] == [str(n) for n in g[f"{tag}.first_names"]]
        np.testing.assert_allclose([tr.Z_MIN, tr.Z_MAX, tr.Z_ABOVE_GROUND_MAX, tr.UVW_MAX, tr.P_MIN, tr.P_MAX],
                                   g[f"{tag}.norms"], rtol=1e-12)
        assert torch.equal(x, T(g[f"{tag}.x"])) and torch.equal(y, T(g[f"{tag}.y"]))
        np.random.seed(77)
        n_draws = 6 if tag == "slice_aug" else 2
        rots = set()
        for i in range(n_draws):
            LR, HR, Z = tr[i]
            for name, t in (("LR", LR), ("HR", HR), ("Z", Z)):
                want = T(g[f"{tag}.train{i}.{name}"])
                assert t.shape == want.shape and t.dtype == torch.float32
                assert torch.allclose(t, want, rtol=1e-6, atol=1e-7), (tag, i, name)
            rots.add(float(HR[0].sum()))
        assert len(rots) == n_draws  # (different augmentations / slices were drawn)
        item = te[0]
        for name, t in zip(("LR", "HR", "Z"), item[:3]):
            assert torch.allclose(t, T(g[f"{tag}.test0.{name}"]), rtol=1e-6, atol=1e-7), (tag, name)
        assert item[3] == str(g[f"{tag}.test0.name"])
        if tag == "interp":
            assert torch.allclose(item[4], T(g["interp.test0.HR_raw"]), rtol=1e-6, atol=1e-7)
            assert torch.allclose(item[5], T(g["interp.test0.Z_raw"]), rtol=1e-6, atol=1e-7)
            back = pd.reverse_interpolate_z_axis(item[1].numpy()[None], item[5].numpy()[None], item[2].numpy()[None])
            assert torch.allclose(back, T(g["interp.test0.HR_back"]), rtol=1e-5, atol=1e-6)
        assert torch.allclose(va[1][0], T(g[f"{tag}.val1.LR"]), rtol=1e-6, atol=1e-7)


def test_preprosess_split_shapes_and_normalisation(data_root):
    from gan_sr_wind_field_amd import process_data as pd

    tr, te, va, x, y = pd.preprosess(Z_DICT=ZD, start_date=date(2018, 3, 1), end_date=date(2018, 3, 3),
                                     include_pressure=True, include_z_channel=True, interpolate_z=True,
                                     enable_slicing=True, slice_size=64, include_above_ground_channel=False,
                                     train_aug_rot=True, train_aug_flip=True, COARSENESS_FACTOR=4)
    assert (len(tr), len(te), len(va)) == (57, 7, 8)  # chronological 80 / 10 / 10 of 72 hourly samples
    assert x.shape == (64,) and x.dtype == torch.float32
    LR, HR, Z = tr[3]
    assert LR.shape == (5, 16, 16, 10) and HR.shape == (3, 64, 64, 10) and Z.shape == (1, 64, 64, 10)
    assert LR.dtype == HR.dtype == Z.dtype == torch.float32
    assert float(HR.abs().max()) <= 1.0 + 1e-6 and 0.0 <= float(LR[3].min()) and float(LR[3].max()) <= 1.0
    assert (Z[0, :, :, 1:] > Z[0, :, :, :-1]).all()
    # normalisation factors come from the training part of the period only and are persisted
    nf = pickle.load(open(data_root / "full_dataset_files" / tr.subfolder_name / "norm_factors.pkl", "rb"))
    assert nf[3] == tr.UVW_MAX and tr.UVW_MAX == te.UVW_MAX
    # test samples: un-augmented full domain + name + raw (un-interpolated) truth
    LRt, HRt, Zt, name, HR_raw, Z_raw = te[0]
    assert HRt.shape == (3, 128, 128, 10) and name.startswith("2018-03-") and HR_raw.shape == HRt.shape
    assert os.path.isfile(data_root / "interpolated_z_data" / tr.subfolder_name / (name + ".pkl"))


def test_interpolation_helpers_match_numpy():
    from gan_sr_wind_field_amd import process_data as pd

    rng = np.random.default_rng(0)
    old = np.cumsum(rng.uniform(1, 50, (6, 5, 10)), axis=-1)
    vals = rng.normal(size=(6, 5, 10))
    new = np.linspace(old[..., 0].mean() - 5, old[..., -1].mean() + 5, 10)
    got = pd._interp_columns(new, old, vals)
    ref = np.stack([[np.interp(new, old[i, j], vals[i, j]) for j in range(5)] for i in range(6)])
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    # flat levels + terrain, and the way back for a field that is linear in z
    terrain = rng.uniform(0, 300, (6, 5))
    lin = 2.0 + 0.01 * old
    z, above, u, *_ = pd.interpolate_z_axis(None, None, old, lin.copy(), lin.copy(), lin.copy(), lin.copy(), terrain)
    assert np.allclose(z - above, terrain[:, :, None]) and np.ptp(above, axis=(0, 1)).max() < 1e-9
    back = pd.reverse_interpolate_z_axis(u[None, None], (old + terrain[:, :, None])[None, None], z[None, None])
    ref_back = np.stack([[np.interp(old[i, j] + terrain[i, j], z[i, j], u[i, j]) for j in range(5)] for i in range(6)])
    np.testing.assert_allclose(back[0, 0].numpy(), ref_back, rtol=1e-12)
    # where a raw level is bracketed by un-clamped interpolated levels the round trip is exact for a linear profile
    col_lo, col_hi = old[..., :1], old[..., -1:]
    ok_new = (above >= col_lo) & (above <= col_hi)  # interpolated levels that were not clamped
    lo_ok = np.where(ok_new, above, np.inf).min(axis=-1, keepdims=True)
    hi_ok = np.where(ok_new, above, -np.inf).max(axis=-1, keepdims=True)
    inside = (old >= lo_ok) & (old <= hi_ok)
    assert inside.sum() > 50
    np.testing.assert_allclose(back[0, 0].numpy()[inside], lin[inside], rtol=1e-9)


def test_augmentation_rotates_vector_components():
    from gan_sr_wind_field_amd.process_data import _rotate_wind

    t = torch.randn(4, 6, 6, 3)
    assert torch.equal(_rotate_wind(_rotate_wind(t, 1), 3), t)
    assert torch.equal(_rotate_wind(_rotate_wind(t, 2), 2), t)
    r = _rotate_wind(t, 1)
    # a uniform flow along +x becomes a uniform flow along +y after a quarter turn of the grid
    flow = torch.zeros(3, 4, 4, 2)
    flow[0] = 1.0
    rf = _rotate_wind(flow, 1)
    assert torch.all(rf[1] == 1.0) and torch.all(rf[0] == 0.0)
    assert torch.equal(r[2], torch.rot90(t[2], 1, [0, 1]))  # w and extra channels only move with the grid


def _patch_oracle_nets(monkeypatch):
    import oracle_nets
    from gan_sr_wind_field_amd.GAN_models import wind_field_GAN_3D as mod

    monkeypatch.setattr(mod, "Generator_3D", oracle_nets.OracleGenerator)
    monkeypatch.setattr(mod, "Discriminator_3D", oracle_nets.OracleDiscriminator)


def _small_cfg(tmp_path):
    from gan_sr_wind_field_amd.config.config import Config

    cfg = Config(LOCAL_INI)
    cfg.is_train, cfg.is_test, cfg.is_use = True, False, False
    cfg.gpu_id, cfg.device = None, torch.device("cpu")
    cfg.name = "unit"
    cfg.env.root_path = str(tmp_path / "out")
    cfg.also_log_to_terminal = False
    cfg.generator.num_features, cfg.generator.num_RRDB, cfg.generator.RDB_growth_chan = 16, 1, 8
    cfg.generator.terrain_number_of_features = 4
    cfg.discriminator.num_features = 4
    cfg.gan_config.start_date, cfg.gan_config.end_date = [2018, 3, 1], [2018, 3, 1]
    cfg.gan_config.number_of_z_layers = 4
    cfg.gan_config.interpolate_z = False
    cfg.dataset_train.num_workers = cfg.dataset_val.num_workers = 0
    cfg.dataset_train.batch_size = 2
    cfg.training.niter, cfg.training.val_period, cfg.training.save_model_period = 4, 2, 3
    cfg.training.d_g_train_period, cfg.training.log_period = 1, 1
    return cfg


def test_train_loop_and_evaluation_artifacts(data_root, tmp_path, monkeypatch):
    import logging

    from gan_sr_wind_field_amd import run as runmod
    from gan_sr_wind_field_amd.test import test as evaluate
    from gan_sr_wind_field_amd.train import train

    _patch_oracle_nets(monkeypatch)
    cfg = _small_cfg(tmp_path)
    assert runmod.safe_setup_env_and_cfg(cfg)
    runmod.save_config(cfg, cfg.env.this_runs_folder)
    runmod.setup_logger(cfg)
    dataset_train, dataset_test, dataset_val, x, y = runmod.prepare_data(cfg)
    assert dataset_train[0][0].shape == (4, 16, 16, 4)  # u, v, w + height channel on the LR grid
    gan = train(cfg, dataset_train, dataset_val, x, y)
    run_dir = cfg.env.this_runs_folder
    assert os.path.isfile(os.path.join(run_dir, "config.ini"))
    for f in ("G_3.pth", "D_3.pth", "state_3.pth"):
        assert os.path.isfile(os.path.join(run_dir, f)), f
    for it in (2, 4):
        imgs = pickle.load(open(os.path.join(run_dir, "images", f"val_imgs__it_{it}.pkl"), "rb"))
        assert set(imgs) == {"HR", "SR", "BC", "LR"} and imgs["SR"].shape == imgs["HR"].shape == (3, 64, 64, 4)
    for h in logging.getLogger("train").handlers:
        h.flush()
    log_text = open(cfg.env.train_log_file).read()
    assert "it 4 " in log_text and "total:" in log_text
    # evaluation harness on the held-out part, from the checkpoint written above
    cfg.is_train, cfg.is_test = False, True
    cfg.env.generator_load_path = os.path.join(run_dir, "G_3.pth")
    avg = evaluate(cfg, dataset_test)
    assert np.isfinite(avg["PSNR"]) and avg["pix"] > 0 and avg["average_wind_speed"] > 0
    rows = open(os.path.join("test_output", "unit____metrics.csv")).read().strip().splitlines()
    assert rows[0].startswith("field,PSNR,PSNR_trilinear") and len(rows) == 1 + len(dataset_test)
    assert os.path.isfile(os.path.join("test_output", "averages.csv"))
    assert any(f.startswith("test_fields_") for f in os.listdir(os.path.join(run_dir, "fields")))


def test_cli_flags_match_reference():
    from gan_sr_wind_field_amd import run as runmod

    cfg = runmod.argv_to_cfg(["--train", "--test", "--slurm_array_id", "3"])
    assert cfg.is_train and cfg.is_test and not cfg.is_use and cfg.slurm_array_id == 3
    with pytest.raises(NotImplementedError):
        runmod.main(["--download"])
    with pytest.raises(NotImplementedError):
        runmod.main(["--param_search"])


def test_validation_figures(tmp_path):
    """the validation epoch's TensorBoard figures (reference train.py:236-307, :340-555): two slices x (comparison,
    error) figure, tags as in the reference; PNG files when no writer is given"""
    from gan_sr_wind_field_amd.tools import valfigures

    if not valfigures.available():
        pytest.skip("matplotlib not installed")
    rng = np.random.default_rng(3)
    hr = rng.normal(size=(3, 16, 16, 6)).astype(np.float32)
    imgs = {"HR": hr, "SR": hr + 0.1 * rng.normal(size=hr.shape).astype(np.float32),
            "BC": hr + 0.3 * rng.normal(size=hr.shape).astype(np.float32), "LR": hr[:, ::4, ::4]}

    class Writer:
        def __init__(self):
            self.calls = []

        def add_figure(self, tag, fig, it):
            self.calls.append((tag, it, len(fig.axes)))

    w = Writer()
    tags = valfigures.log_validation_figures(imgs, 40, tb=w, rng=np.random.default_rng(0))
    assert [c[0] for c in w.calls] == tags and len(tags) == 4 and all(c[1] == 40 for c in w.calls)
    assert tags[0] == "im/40/wind_fields/u_field_z_index3" and tags[1] == "im/40/Error/u_field_z_index3"
    assert tags[2].startswith("im/40/wind_fields/") and tags[3].startswith("im/40/Error/")
    assert w.calls[0][2] == 5 and w.calls[1][2] == 12  # 4 panels + 1 colour bar; 6 panels + 6 colour bars
    out = tmp_path / "images"
    tags2 = valfigures.log_validation_figures(imgs, 41, out_dir=str(out), rng=np.random.default_rng(0))
    assert len(tags2) == 4 and len(list(out.glob("*.png"))) == 4
    assert valfigures.log_validation_figures(imgs, 42) == []  # nowhere to send them
