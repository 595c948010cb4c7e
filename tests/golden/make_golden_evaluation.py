#!/usr/bin/env python3
"""Golden vectors for the 2-D evaluation loop, produced BY THE REFERENCE.

/root/reference/evaluation/{evaluation_utils,renderer}.py cannot be imported here (cv2, open3d, lpips are not
installed), but the functions pinned below are plain numpy / torch: their definitions are parsed out of the
reference files and executed AT GENERATION TIME (nothing is copied into the repository):
  evaluation_utils.py: calculate_depth_metrics_2d (:380-415), calculate_mssim (:321-377)
  renderer.py: Renderer._calculate_pred2gt_transformation (:79-111),
               Renderer.transform_camera_extrinsics_gt2pred (:276-287), Renderer.transform_matrices_pred2gt (:289-298)
Writes tests/golden/evaluation_golden.npz.   python tests/golden/make_golden_evaluation.py
"""
import ast
import os
import types

import numpy as np
import torch
import tqdm


def _functions(path, names, cls=None):
    tree = ast.parse(open(path).read())
    body = tree.body if cls is None else next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    picked = [n for n in body if isinstance(n, ast.FunctionDef) and n.name in names]
    for fn in picked:
        fn.decorator_list = []  # staticmethods become plain functions
    ns = {"np": np, "torch": torch, "tqdm": tqdm}
    exec(compile(ast.Module(body=picked, type_ignores=[]), os.path.basename(path), "exec"), ns)
    return ns


def _pose(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    m = np.eye(4)
    m[:3, :3] = q
    m[:3, 3] = rng.normal(size=3)
    return m


def build():
    utils = _functions("/root/reference/evaluation/evaluation_utils.py", {"calculate_depth_metrics_2d", "calculate_mssim"})
    rend = _functions("/root/reference/evaluation/renderer.py",
                      {"_calculate_pred2gt_transformation", "transform_camera_extrinsics_gt2pred",
                       "transform_matrices_pred2gt"}, cls="Renderer")
    rng = np.random.default_rng(11)
    out = {}

    # ---- depth metrics: valid-range masking on both sides, with and without the scale alignment
    for i, (shape, scale) in enumerate((((24, 32), 1.0), ((17, 9), 0.6), ((40, 40), 1.7))):
        gt = rng.uniform(-0.3, 6.0, shape)
        pred = gt / scale * (1 + 0.08 * rng.normal(size=shape)) + 0.02 * rng.normal(size=shape)
        out[f"depth{i}_gt"], out[f"depth{i}_pred"] = gt, pred
        for with_scale in (True, False):
            m = utils["calculate_depth_metrics_2d"](gt.copy(), pred.copy(), with_scale=with_scale)
            out[f"depth{i}_metrics_scale{int(with_scale)}"] = np.array([m[k] for k in sorted(m)])
    out["depth_metric_names"] = np.array(sorted(m))

    # ---- MSSIM on [-1, 1] images (what calculate_color_metrics_2d feeds it)
    for i, (h, w, noise) in enumerate(((24, 32, 0.05), (13, 40, 0.4), (16, 16, 0.0))):
        a = rng.uniform(-1, 1, (1, 3, h, w)).astype(np.float32)
        b = np.clip(a + noise * rng.normal(size=a.shape), -1, 1).astype(np.float32)
        out[f"ssim{i}_a"], out[f"ssim{i}_b"] = a, b
        out[f"ssim{i}_value"] = np.array(utils["calculate_mssim"](torch.tensor(a), torch.tensor(b)))

    # ---- pred<->gt alignment from 5 keyframes with per-frame scale jitter and out-of-range pixels
    n_kf, shape = 5, (12, 16)
    true_scale = 2.3
    depths_gt = [rng.uniform(0.2, 5.5, shape) for _ in range(n_kf)]
    depths_pred = [g / (true_scale * (1 + 0.05 * rng.normal())) + 0.01 * rng.normal(size=shape) for g in depths_gt]
    depths_pred[2][:3] = -1.0
    extr_gt = np.stack([_pose(rng) for _ in range(7)])
    extr_pred0 = _pose(rng)
    fake = types.SimpleNamespace(
        keyframes=list(range(n_kf)),
        dataset=types.SimpleNamespace(camera_intrinsics={}, camera_extrinsics=extr_gt,
                                      frames_depth=lambda mode, keyframes: depths_gt),
        nerf=types.SimpleNamespace(
            render_frame_depth_from_training_frame=lambda camera_intrinsics, frame_index: depths_pred[frame_index].copy(),
            get_camera_extrinsics=lambda frame_index: extr_pred0.copy()))
    rend["_calculate_pred2gt_transformation"](fake)
    tf = fake.pred2gt_transformation
    out["align_depths_gt"], out["align_depths_pred"] = np.stack(depths_gt), np.stack(depths_pred)
    out["align_extr_gt"], out["align_extr_pred0"] = extr_gt, extr_pred0
    out["align_scale"] = np.array(tf["scale_pred2gt"])
    out["align_matrix"], out["align_matrix_scaled"] = tf["matrix_pred2gt"], tf["matrix_pred2gt_scaled"]
    out["align_gt2pred"] = rend["transform_camera_extrinsics_gt2pred"](extr_gt, tf)
    out["align_pred2gt"] = rend["transform_matrices_pred2gt"](out["align_gt2pred"], tf)
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "evaluation_golden.npz")
    np.savez_compressed(path, **build())
    print("wrote", path, os.path.getsize(path), "bytes")
