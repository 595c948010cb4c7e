"""GPU half of tests/test_dataset_golden_cpu.py: the keyframe buffer on the MI355X against the vectors the reference's
own DynamicDataset methods produced.  Indexing / permutes are exact on any device; the two linear solves (world
normalisation, world-space normals) go through the GPU's LAPACK: 1e-6."""
import numpy as np
import pytest
import torch

from test_dataset_golden_cpu import CASES, GOLDEN, _packet

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("schedule,normals", CASES, ids=[f"{s}-normals{n}" for s, n in CASES])
def test_ingest_matches_the_reference_on_the_gpu(device, schedule, normals):
    from nerf_vo_amd.mapping.dataset import DynamicDataset

    g = np.load(GOLDEN)
    tag = f"{schedule}_n{normals}"
    ds = DynamicDataset(num_frames=int(g["num_frames"]), frame_height=int(g["height"]), frame_width=int(g["width"]),
                        device=device, use_normals=bool(normals))
    for i in range(int(g[f"{tag}_packets"])):
        ds.update(_packet(g, tag, i, device))
        assert ds.num_active_frames == int(g[f"{tag}_p{i}_num_active"])
        for buf in ("camera_intrinsics", "frames_color", "frames_depth") + (("frames_normal",) if normals else ()):
            assert np.array_equal(getattr(ds, buf).cpu().numpy(), g[f"{tag}_p{i}_{buf}"]), buf
        np.testing.assert_allclose(ds.camera_extrinsics.cpu().numpy(), g[f"{tag}_p{i}_camera_extrinsics"], rtol=1e-5, atol=2e-6)
        if normals:
            np.testing.assert_allclose(ds.get_dataset()["normal_image"].cpu().numpy(), g[f"{tag}_p{i}_ds_normal_image"],
                                       rtol=0, atol=2e-6)
