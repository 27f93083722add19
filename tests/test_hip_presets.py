"""Every BASELINE.json configuration is executed on the GPU (``bench.py --config``):

  C1    shipped local ini: D with slicing, LR 16x16x10 -> HR 64x64x10 (parity: test_hip_fullsize_parity.py)
  C1b   the reference's real patch size 32x32x10 -> 128x128x10, no slicing
  C1c   the reference's production training shape: cluster ini as shipped (batch 32, D sliced, 64x64x10 HR slices;
        config/wind_field_GAN_3D_config_cluster.ini:42,47)
  C2    generator-only fwd + bwd + Adam, fp32, 64x64x64 -> 256x256x64
  C3'   the benchmark default (full-size properties: test_hip_networks.py::test_full_size_c3_*)
  C3lit BASELINE.json configs[2] read literally: generator-only at LR 128^3 -> 512 x 512 x 128 (144-channel HR tensors
        of 4.8e9 elements: beyond 32-bit element offsets; D cannot consume 512 x 512)
  C4    the per-GPU shape of the 8-GPU run: C3' at batch 4
  C5b   upscale8 ini (x8, three UpConv stages), batch 8, 16x16x10 -> 128x128x10, full G + D step
  C6    upscale16 ini (x16, four UpConv stages; pretrained_models/upscale16_pix4_no_adv_no_slicing/config.ini:5),
        batch 8, 8x8x10 -> 128x128x10, full G + D step
  C5lit x8 generator-only at 64x64x64 -> 512x512x64 (4.8 GB per 144-channel HR tensor: the HBM stress case)

Per preset, size-independent properties: one step runs with finite losses and moves the weights, the eval
forward is deterministic (bit-identical twice), and the bf16 output sits at bf16 distance from the fp32 program
on the same weights (the fp32 program is pinned to the reference elsewhere).
"""
import os
import sys
from types import SimpleNamespace

import pytest
import torch

from conftest import REPO, rel_l2

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

sys.path.insert(0, REPO)


def _bench():
    import importlib.util

    spec = importlib.util.spec_from_file_location("wsr_bench", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


@pytest.mark.parametrize("preset", ["C1", "C1b", "C1c", "C2", "C3lit", "C4", "C5b", "C5lit", "C6"])
def test_preset_runs(hip, preset):
    from gan_sr_wind_field_amd.process_data import synthetic_batch

    bench = _bench()
    ini, n, nz, B, dtype, kind, slicing, _ = bench.PRESETS[preset]
    args = SimpleNamespace(ini=ini, slicing=slicing, nz=nz)
    gan, cfg = bench.make_gan(args, DEV, dtype)
    s = cfg.scale
    LR, HR, Z, x, y = (t.to(DEV) for t in synthetic_batch(B, n, nz, s, seed=11))
    gan.feed_xy_niter(x, y, torch.tensor(cfg.training.niter, device=DEV), 1, 1)
    w0 = gan.G.hr_convs[2].weight.detach().clone()
    if kind == "gan":
        gan.optimize_parameters(LR, HR, Z, 0)
        gan.optimize_parameters(LR, HR, Z, 1)
        for d in (gan.get_G_train_loss_dict_ref(), gan.get_D_loss_dict_ref()):
            assert all(torch.isfinite(v.detach()).all() for k, v in d.items() if "validation" not in k), preset
    else:
        gan.G.train()
        gan.optimizer_G.zero_grad(set_to_none=True)
        loss = torch.nn.functional.l1_loss(gan.G(LR, Z), HR)
        loss.backward()
        assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in gan.G.parameters())
        gan.optimizer_G.step()
    assert not torch.equal(w0, gan.G.hr_convs[2].weight)
    gan.G.eval()
    with torch.no_grad():
        a = gan.G(LR, Z)
        assert a.shape == (B, 3, s * n, s * n, nz) and torch.isfinite(a).all()
        assert torch.equal(a, gan.G(LR, Z))
        # the fp32 program on the same weights - also for C5lit / C3lit, whose 144-channel HR tensors hold 2.4e9 / 4.8e9
        # elements (19 GB in fp32 at C3lit): every kernel of the fp32 program indexes with 64-bit element offsets, the
        # bf16 tile kernels with 32-bit offsets relative to the halo's first x-plane (64-bit workgroup base)
        if dtype == "bf16":
            sd = gan.G.state_dict()
            del gan
            torch.cuda.empty_cache()
            gan32, _ = bench.make_gan(args, DEV, "fp32")
            gan32.G.load_state_dict(sd)
            gan32.G.eval()
            b = gan32.G(LR, Z)
            assert rel_l2(a, b) < 3e-2, rel_l2(a, b)
    torch.cuda.empty_cache()
