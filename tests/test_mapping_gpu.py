"""GPU end-to-end: the mapper mirror (Nerfstudio.__call__/update/train/shut_down/save_snapshot) fed like
the reference's MappingModule feeds it, then NerfstudioRenderer + both PSNR definitions."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_synthetic_mapping_end_to_end(device, tmp_path):
    from run_synthetic_mapping import run

    res = run(keyframes=16, height=120, width=160, iterations=400, eval_frames=3, chunk=8, quiet=True,
              out_dir=str(tmp_path))
    assert np.isfinite(res["psnr_float_mse"]) and res["psnr_float_mse"] > 17.0, res
    assert res["depth_l1"] < 1.0, res
    # snapshot artefacts of /root/reference/nerf_vo/mapping/nerfstudio.py:198-217
    assert (tmp_path / "dataset.pt").exists()
    mats = json.load(open(tmp_path / "matrices" / "matrices_origin2frame_training.json"))
    assert np.asarray(mats).shape == (16, 4, 4)
    assert list((tmp_path / "snapshots").glob("step-*.ckpt")) or list(tmp_path.rglob("step-*.ckpt"))


def test_psnr_definitions():
    from nerf_vo_amd.mapping.renderer import calculate_psnr_float, calculate_psnr_reference
    from oracle.rays import psnr_float, psnr_reference

    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-40, 41, a.shape), 0, 255).astype(np.uint8)
    assert calculate_psnr_reference(a, b) == pytest.approx(psnr_reference(a, b))
    assert calculate_psnr_float(a, b) == pytest.approx(psnr_float(a, b))
    # the reference's uint8 arithmetic wraps: true MSE 5050 vs wrapped "MSE" 58
    x = np.zeros((1, 2, 3), np.uint8)
    y = np.zeros((1, 2, 3), np.uint8)
    y[0, 0] = 100
    y[0, 1] = 10
    assert np.mean((x[..., 0] - y[..., 0]) ** 2) == 58.0
    assert calculate_psnr_reference(x, y) > calculate_psnr_float(x, y)
