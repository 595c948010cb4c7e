#!/usr/bin/env python3
"""Golden vectors for the keyframe buffer of the mapping stage, produced BY THE
REFERENCE's own methods.

/root/reference/nerf_vo/mapping/nerfstudio_utils.py cannot be imported (its top-level imports resolve into the empty
nerfstudio submodule), but the methods of ``DynamicDataset`` that make up the ingest path -- ``update`` / ``prepare_update``
/ ``insert_update`` (:157-228), ``get_dataset`` / ``get_frame`` / ``__getitem__`` / ``__len__`` (:110-155) and ``save_dataset``
(:230-241) -- are plain torch.  Their definitions are parsed out of the reference file and executed AT GENERATION TIME
(nothing is copied into the repository), bound to an ``object.__new__`` instance that carries the attributes and
pre-allocated buffers of the constructor (:37-74: zeros / tiled identity, fp32) -- the constructor itself is not run: it
builds nerfstudio SceneBox / Cameras objects, which do not exist here.

Two ingest schedules are driven through them, each with and without normals:
  * "droid": every pose comes with its frame (#extrinsics == #colours): slots addressed by keyframe_indices, a later
    packet overwrites some of them;
  * "dpvo":  new frames are appended while poses and depths of the whole sliding window are refreshed
    (#extrinsics != #colours), the shape NeRF-VO's own configs produce (nerf_vo/tracking/dpvo.py:85-99).
Stored: every input packet and, after every packet, the resulting buffers, num_active_frames, len(), the normalisation
matrix, get_dataset() (incl. the per-step normal solve) and two get_frame() rows; the final save_dataset() file's
tensors.  Writes tests/golden/dataset_golden.npz.   python tests/golden/make_golden_dataset.py
"""
import ast
import os
import tempfile

import numpy as np
import torch

REF = "/root/reference/nerf_vo/mapping/nerfstudio_utils.py"
NUM_FRAMES, H, W = 10, 6, 8


def reference_class():
    tree = ast.parse(open(REF).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "DynamicDataset")
    cls.body = [n for n in cls.body if not (isinstance(n, ast.FunctionDef) and n.name == "__init__")]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), os.path.basename(REF), "exec"), ns)
    return ns["DynamicDataset"]


def reference_instance(cls, normals: bool):
    """Attributes and buffers as the reference's constructor leaves them (nerfstudio_utils.py:37-74)."""
    ds = object.__new__(cls)
    dev = torch.device("cpu")
    ds.device, ds.use_normals, ds.num_frames, ds.num_active_frames = dev, normals, NUM_FRAMES, 0
    ds.frame_height, ds.frame_width, ds.normalization_matrix = H, W, None
    ds.camera_intrinsics = torch.zeros((NUM_FRAMES, 4), dtype=torch.float32, device=dev)
    ds.camera_extrinsics = torch.tile(torch.eye(4, dtype=torch.float32, device=dev), (NUM_FRAMES, 1, 1))
    ds.frames_color = torch.zeros((NUM_FRAMES, H, W, 3), dtype=torch.float32, device=dev)
    ds.frames_depth = torch.zeros((NUM_FRAMES, H, W, 1), dtype=torch.float32, device=dev)
    if normals:
        ds.frames_normal = torch.zeros((NUM_FRAMES, H, W, 3), dtype=torch.float32, device=dev)
    return ds


def pose(g):
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    if torch.det(q) < 0:
        q[:, 0] *= -1
    m = torch.eye(4)
    m[:3, :3] = q
    m[:3, 3] = torch.randn(3, generator=g)
    return m


def packet(g, key_idx, n_new, normals):
    k = len(key_idx)
    p = {"keyframe_indices": torch.tensor(key_idx, dtype=torch.long),
         "camera_intrinsics": torch.rand(n_new, 4, generator=g) * 100 + 200,
         "camera_extrinsics": torch.stack([pose(g) for _ in range(k)]),
         "frames_color": torch.rand(n_new, 3, H, W, generator=g),
         "frames_depth": torch.rand(k, 1, H, W, generator=g) * 5}
    if normals:
        p["frames_normal"] = torch.nn.functional.normalize(torch.randn(n_new, 3, H, W, generator=g), dim=1)
    return p


SCHEDULES = {
    # (keyframe_indices, number of NEW frames in the packet)
    "droid": [([0, 1, 2], 3), ([3, 4], 2), ([1, 4, 5], 3)],
    "dpvo": [([0, 1, 2], 3), ([0, 1, 2, 3, 4], 2), ([2, 3, 4, 5], 1), ([3, 4, 5, 6, 7], 2)],
}


def build():
    cls = reference_class()
    out = {"num_frames": np.array(NUM_FRAMES), "height": np.array(H), "width": np.array(W)}
    for name, sched in SCHEDULES.items():
        for normals in (False, True):
            tag = f"{name}_n{int(normals)}"
            g = torch.Generator().manual_seed({"droid": 11, "dpvo": 23}[name] + int(normals))
            ds = reference_instance(cls, normals)
            out[f"{tag}_packets"] = np.array(len(sched))
            for i, (key_idx, n_new) in enumerate(sched):
                p = packet(g, key_idx, n_new, normals)
                for k, v in p.items():
                    out[f"{tag}_p{i}_in_{k}"] = v.numpy().copy()
                ds.update(input=p)
                n = ds.num_active_frames
                out[f"{tag}_p{i}_num_active"] = np.array(n)
                out[f"{tag}_p{i}_len"] = np.array(len(ds))
                out[f"{tag}_p{i}_normalization"] = ds.normalization_matrix.numpy().copy()
                for buf in ("camera_intrinsics", "camera_extrinsics", "frames_color", "frames_depth") + (
                        ("frames_normal",) if normals else ()):
                    out[f"{tag}_p{i}_{buf}"] = getattr(ds, buf).numpy().copy()
                data = ds.get_dataset()
                out[f"{tag}_p{i}_ds_image_idx"] = data["image_idx"].numpy().copy()
                out[f"{tag}_p{i}_ds_image"] = data["image"].numpy().copy()
                out[f"{tag}_p{i}_ds_depth_image"] = data["depth_image"].numpy().copy()
                if normals:
                    out[f"{tag}_p{i}_ds_normal_image"] = data["normal_image"].numpy().copy()
                    for f in (0, n - 1):
                        out[f"{tag}_p{i}_frame{f}_normal_image"] = ds[f]["normal_image"].numpy().copy()
            with tempfile.TemporaryDirectory() as tmp:
                ds.save_dataset(dir_prediction=tmp)
                saved = torch.load(f"{tmp}/dataset.pt")
            out[f"{tag}_saved_keys"] = np.array(sorted(saved))
            for k, v in saved.items():
                out[f"{tag}_saved_{k}"] = v.numpy().copy()
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dataset_golden.npz")
    np.savez_compressed(path, **build())
    print("wrote", path, os.path.getsize(path), "bytes")
