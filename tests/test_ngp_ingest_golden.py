"""Ingest of the occupancy-grid back-end (nerf_vo_amd/mapping/instant_ngp_mapper.py::InstantNGP.update) against what
the REFERENCE's own ``InstantNGP.update`` (/root/reference/nerf_vo/mapping/instant_ngp.py:61-102) hands to
``update_training_images`` -- tests/golden/make_golden_ngp_ingest.py runs that method against a recording testbed.
Row a13 (ingest half) of SURVEY.md section 8, pinned by the reference.  The mirror passes device tensors where the
reference passes lists of host arrays (the facade accepts both); values, order, shapes and the scalar arguments must be
the reference's -- bit for bit on the CPU, to the accuracy of the device's pow (2e-6 relative) on the GPU."""
import argparse
import os
import types

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ngp_ingest_golden.npz")


def _drive(device):
    from nerf_vo_amd.mapping.instant_ngp_mapper import InstantNGP

    g = np.load(GOLDEN)
    got = []
    for i in range(2):
        received = {}
        self = object.__new__(InstantNGP)
        self.device = torch.device(device)
        self.args = argparse.Namespace(frame_width=int(g["width"]), frame_height=int(g["height"]))
        self.is_initialized = False
        self.ngp = types.SimpleNamespace(nerf=types.SimpleNamespace(training=types.SimpleNamespace(
            update_training_images=lambda **kw: received.update(kw))))
        packet = {k[len(f"p{i}_in_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"p{i}_in_")}
        self.update(input=packet)
        assert self.is_initialized is True
        got.append(received)
    return g, got


def _check(g, got, exact: bool):
    for i, rec in enumerate(got):
        assert sorted(rec) == [str(k) for k in g[f"p{i}_kwargs"]], "argument names of update_training_images"
        assert list(rec["frame_ids"]) == g[f"p{i}_frame_ids"].tolist()
        for name in ("poses", "images", "depths", "depths_cov"):
            mine = rec[name]
            mine = (torch.stack(list(mine)) if isinstance(mine, (list, tuple)) else mine).detach().cpu().numpy()
            ref = g[f"p{i}_{name}"]
            assert mine.shape == ref.shape and mine.dtype == ref.dtype, f"{name}: {mine.shape} {mine.dtype} vs {ref.shape} {ref.dtype}"
            if exact:
                assert np.array_equal(mine, ref), f"packet {i}: {name} differs from what the reference hands to the testbed"
            else:
                np.testing.assert_allclose(mine, ref, rtol=2e-6, atol=1e-8, err_msg=f"packet {i}: {name}")
        for name in ("resolution", "principal_point", "focal_length"):
            assert np.array_equal(np.asarray(rec[name]), g[f"p{i}_{name}"]), name
        assert float(rec["depth_scale"]) == float(g[f"p{i}_depth_scale"]) == 1.0
        assert float(rec["depth_cov_scale"]) == float(g[f"p{i}_depth_cov_scale"]) == 1.0
        # the sRGB knee: 0 -> 0, 0.04045 -> the linear branch, 1 -> 1; alpha is one everywhere
        img = np.asarray(g[f"p{i}_images"])
        assert img.shape[-1] == 4 and (img[..., 3] == 1.0).all()


def test_ngp_ingest_matches_the_reference_cpu():
    g, got = _drive("cpu")
    _check(g, got, exact=True)


@pytest.mark.gpu
def test_ngp_ingest_matches_the_reference_gpu(device):
    g, got = _drive(device)
    _check(g, got, exact=False)
