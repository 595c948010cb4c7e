"""The keyframe buffer mirror (nerf_vo_amd/mapping/dataset.py) against vectors the REFERENCE's own DynamicDataset methods
produced (tests/golden/make_golden_dataset.py parses update / prepare_update / insert_update / get_dataset / get_frame /
save_dataset out of /root/reference/nerf_vo/mapping/nerfstudio_utils.py:110-241 and runs them; the reference itself is
not needed here).  Rows a2 / a15 of SURVEY.md section 8: pinned by the reference, not by a reading of it.

Bit-for-bit on the CPU: ingest is indexing, permutes and two small linear solves -- the mirror issues the same torch
calls on the same values, so every buffer, the normalisation matrix, the solved normal images and the saved dataset file
must be identical to the last bit."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_golden.npz")
CASES = [(s, n) for s in ("droid", "dpvo") for n in (0, 1)]


def _packet(g, tag, i, device="cpu"):
    keys = ["keyframe_indices", "camera_intrinsics", "camera_extrinsics", "frames_color", "frames_depth", "frames_normal"]
    return {k: torch.from_numpy(g[f"{tag}_p{i}_in_{k}"]).to(device) for k in keys if f"{tag}_p{i}_in_{k}" in g}


def _same(a: torch.Tensor, ref: np.ndarray) -> bool:
    return a.dtype == torch.from_numpy(ref).dtype and np.array_equal(a.detach().cpu().numpy(), ref)


@pytest.mark.parametrize("schedule,normals", CASES, ids=[f"{s}-normals{n}" for s, n in CASES])
def test_ingest_matches_the_reference_bit_for_bit(schedule, normals, tmp_path):
    from nerf_vo_amd.mapping.dataset import DynamicDataset

    g = np.load(GOLDEN)
    tag = f"{schedule}_n{normals}"
    ds = DynamicDataset(num_frames=int(g["num_frames"]), frame_height=int(g["height"]), frame_width=int(g["width"]),
                        device=torch.device("cpu"), use_normals=bool(normals))
    for i in range(int(g[f"{tag}_packets"])):
        ds.update(_packet(g, tag, i))
        n = int(g[f"{tag}_p{i}_num_active"])
        assert ds.num_active_frames == n and len(ds) == int(g[f"{tag}_p{i}_len"])
        assert _same(ds.normalization_matrix, g[f"{tag}_p{i}_normalization"]), "world normalisation matrix"
        for buf in ("camera_intrinsics", "camera_extrinsics", "frames_color", "frames_depth") + (("frames_normal",) if normals else ()):
            assert _same(getattr(ds, buf), g[f"{tag}_p{i}_{buf}"]), f"packet {i}: buffer {buf} differs from the reference's"
        data = ds.get_dataset()
        assert _same(data["image_idx"], g[f"{tag}_p{i}_ds_image_idx"])
        assert _same(data["image"], g[f"{tag}_p{i}_ds_image"]) and _same(data["depth_image"], g[f"{tag}_p{i}_ds_depth_image"])
        if normals:
            # the reference re-solves every active frame's normals on EVERY get_dataset() (nerfstudio_utils.py:145-153);
            # the mirror solves at ingest and caches: same values, bit for bit, including after a pose refresh
            assert _same(data["normal_image"], g[f"{tag}_p{i}_ds_normal_image"]), f"packet {i}: world normals"
            for f in (0, n - 1):
                assert _same(ds[f]["normal_image"], g[f"{tag}_p{i}_frame{f}_normal_image"])
        # aliasing contract (nerfstudio_utils.py:90-107): Cameras holds views of the buffers
        assert ds.cameras.camera_to_worlds.data_ptr() == ds.camera_extrinsics.data_ptr()
    ds.save_dataset(dir_prediction=str(tmp_path))
    saved = torch.load(tmp_path / "dataset.pt")
    assert sorted(saved) == [str(k) for k in g[f"{tag}_saved_keys"]]
    for k, v in saved.items():
        assert _same(v, g[f"{tag}_saved_{k}"]), f"dataset.pt[{k}]"
    # and the file loads back into a dataset whose buffers equal the reference's (dir_prediction path, :76-88)
    again = DynamicDataset(num_frames=int(g["num_frames"]), frame_height=int(g["height"]), frame_width=int(g["width"]),
                           device=torch.device("cpu"), use_normals=bool(normals), dir_prediction=str(tmp_path))
    assert again.num_active_frames == ds.num_active_frames
    assert torch.equal(again.frames_color[: ds.num_active_frames], ds.frames_color[: ds.num_active_frames])
    assert torch.equal(again.camera_extrinsics[: ds.num_active_frames], ds.camera_extrinsics[: ds.num_active_frames])
