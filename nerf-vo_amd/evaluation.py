"""2-D evaluation loop around the native renderer: prediction<->ground-truth
alignment, rendering of evaluation frames to disk, and the depth / colour metrics computed from those files.

Mirrors the behaviour of the reference's
  * ``Renderer._calculate_pred2gt_transformation`` / ``transform_camera_extrinsics_gt2pred`` /
    ``transform_matrices_pred2gt`` / ``render_frames`` / ``_render_frame``
    (/root/reference/evaluation/renderer.py:79-124, 239-298),
  * ``calculate_depth_metrics_2d`` / ``calculate_psnr_color`` / ``calculate_mssim`` /
    ``calculate_color_metrics_2d`` (/root/reference/evaluation/evaluation_utils.py:289-443),
  * ``Evaluator.calculate_metrics_2d`` (/root/reference/evaluation/evaluator.py:88-146),
over ``nerf_vo_amd.mapping.renderer.NeRFRenderer`` objects.  Images are written with PIL (JPEG quality 95 is
OpenCV's ``imwrite`` default; 16-bit PNG depth) because OpenCV is not part of this image.  LPIPS needs
pretrained AlexNet weights that cannot be fetched here: ``lpips_loss`` is an optional callable and the
``lpips`` column is omitted without it.  3-D (mesh) metrics are out of scope.

The numeric functions are pinned by the reference's own outputs: tests/golden/make_golden_evaluation.py executes
the reference's definitions on seeded inputs and tests/test_evaluation_cpu.py compares.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from .mapping.renderer import calculate_psnr_reference

DEPTH_VALID_MAX = 5.0  # metres; both the alignment and the metrics ignore depths outside (0, 5)


def _valid(depth_gt: np.ndarray, depth_pred: np.ndarray) -> np.ndarray:
    return (depth_gt > 0) & (depth_pred > 0) & (depth_gt < DEPTH_VALID_MAX) & (depth_pred < DEPTH_VALID_MAX)


def depth_scale_pred2gt(depth_gt: np.ndarray, depth_pred: np.ndarray) -> float:
    """Ratio of mean ground-truth to mean predicted depth over jointly valid pixels (renderer.py:88-93)."""
    m = _valid(depth_gt, depth_pred)
    return depth_gt[m].mean() / depth_pred[m].mean()


def estimate_pred2gt(depths_gt, depths_pred, extrinsics_gt0: np.ndarray, extrinsics_pred0: np.ndarray) -> dict:
    """Median per-keyframe depth scale + the rigid / scaled transforms anchored on frame 0
    (renderer.py:79-111).  Returns the reference's dictionary."""
    scale = np.median([depth_scale_pred2gt(g, p) for g, p in zip(depths_gt, depths_pred)])
    inv_pred0 = np.linalg.inv(extrinsics_pred0)
    return {
        "scale_pred2gt": scale,
        "matrix_pred2gt": extrinsics_gt0 @ inv_pred0,
        "matrix_pred2gt_scaled": extrinsics_gt0 @ np.diag([scale, scale, scale, 1]) @ inv_pred0,
    }


def _compose(rotation_from: np.ndarray, translation_from: np.ndarray, extrinsics: np.ndarray) -> np.ndarray:
    """Rotation block taken from ``rotation_from @ E``, translation column from ``translation_from @ E``.
    (The reference spells M @ E as ``(M @ E.T).T`` on the [n,4,4] stack, which is the same product.)"""
    extrinsics = np.asarray(extrinsics)
    out = np.tile(np.eye(4), (extrinsics.shape[0], 1, 1))
    out[:, :3, 3] = (translation_from @ extrinsics)[:, :3, 3]
    out[:, :3, :3] = (rotation_from @ extrinsics)[:, :3, :3]
    return out


def transform_camera_extrinsics_gt2pred(camera_extrinsics: np.ndarray, pred2gt_transformation: dict) -> np.ndarray:
    """Ground-truth poses -> the model's (normalised, unscaled) world (renderer.py:276-287)."""
    return _compose(np.linalg.inv(pred2gt_transformation["matrix_pred2gt"]),
                    np.linalg.inv(pred2gt_transformation["matrix_pred2gt_scaled"]), camera_extrinsics)


def transform_matrices_pred2gt(camera_extrinsics: np.ndarray, pred2gt_transformation: dict) -> np.ndarray:
    """Model-world poses -> ground-truth world (renderer.py:289-298)."""
    return _compose(pred2gt_transformation["matrix_pred2gt"], pred2gt_transformation["matrix_pred2gt_scaled"],
                    camera_extrinsics)


def calculate_depth_metrics_2d(frame_depth_gt: np.ndarray, frame_depth_pred: np.ndarray, with_scale: bool = True) -> dict:
    """evaluation_utils.py:380-415.  Relative errors are relative to the PREDICTION, like the reference."""
    m = _valid(frame_depth_gt, frame_depth_pred)
    gt = frame_depth_gt[m]
    pred = frame_depth_pred[m]
    if with_scale:
        pred = pred * (gt.mean() / pred.mean())
    diff = np.abs(pred - gt)
    ratio = np.maximum(gt / pred, pred / gt)
    return {
        "absolute_relative": np.mean(diff / pred),
        "absolute_difference": np.mean(diff),
        "square_relative": np.mean(diff ** 2 / pred),
        "square_difference": np.sqrt(np.mean(diff ** 2)),
        "square_log_difference": np.sqrt(np.mean((np.log(pred) - np.log(gt)) ** 2)),
        "delta1": np.mean((ratio < 1.25).astype("float")),
        "delta2": np.mean((ratio < 1.25 ** 2).astype("float")),
        "delta3": np.mean((ratio < 1.25 ** 3).astype("float")),
    }


def calculate_mssim(image1: torch.Tensor, image2: torch.Tensor) -> float:
    """Mean SSIM of two [1,3,H,W] tensors with an 11-tap Gaussian window (sigma 1.5), zero padding,
    C1 = 0.01^2, C2 = 0.03^2 (evaluation_utils.py:321-377).  The window is applied as two 1-D passes
    (same operator as the reference's 11x11 depth-wise convolution)."""
    taps, sigma = 11, 1.5
    x = torch.arange(taps, dtype=torch.float32) - taps // 2
    g = torch.exp(-(x ** 2) / (2 * sigma ** 2))
    g = g / g.sum()
    channels = image1.shape[1]
    row = g.view(1, 1, 1, taps).expand(channels, 1, 1, taps).contiguous()
    col = g.view(1, 1, taps, 1).expand(channels, 1, taps, 1).contiguous()

    def blur(t: torch.Tensor) -> torch.Tensor:
        t = torch.nn.functional.conv2d(t, row, padding=(0, taps // 2), groups=channels)
        return torch.nn.functional.conv2d(t, col, padding=(taps // 2, 0), groups=channels)

    with torch.no_grad():
        a, b = image1.float(), image2.float()
        mu_a, mu_b = blur(a), blur(b)
        var_a = blur(a * a) - mu_a ** 2
        var_b = blur(b * b) - mu_b ** 2
        cov = blur(a * b) - mu_a * mu_b
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        ssim = ((2 * mu_a * mu_b + c1) * (2 * cov + c2)) / ((mu_a ** 2 + mu_b ** 2 + c1) * (var_a + var_b + c2))
        return ssim.mean().item()


def calculate_color_metrics_2d(frame_color_gt: np.ndarray, frame_color_pred: np.ndarray, lpips_loss=None) -> dict:
    """PSNR (the reference's uint8-wrapping per-channel definition), MSSIM on [-1,1]-scaled images and --
    when a callable is supplied -- LPIPS on the twice-rescaled tensors the reference feeds it
    (evaluation_utils.py:418-443)."""
    def to_tensor(img: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).unsqueeze(0).float().div(255.0) * 2 - 1

    gt, pred = to_tensor(frame_color_gt), to_tensor(frame_color_pred)
    out = {"psnr": calculate_psnr_reference(frame_color_gt, frame_color_pred), "mssim": calculate_mssim(gt, pred)}
    if lpips_loss is not None:
        out["lpips"] = float(lpips_loss(gt * 2 - 1, pred * 2 - 1))
    return out


# ---------------------------------------------------------------------------------------------------------------
# rendering of evaluation frames to disk
# ---------------------------------------------------------------------------------------------------------------
def write_color_jpeg(path: str, rgb: np.ndarray) -> None:
    from PIL import Image

    Image.fromarray(rgb, mode="RGB").save(path, format="JPEG", quality=95)


def read_color(path: str) -> np.ndarray:
    from PIL import Image

    return np.asarray(Image.open(path).convert("RGB"))


def write_depth_png16(path: str, depth_units: np.ndarray) -> None:
    from PIL import Image

    Image.fromarray(depth_units.astype(np.uint16)).save(path, format="PNG")


def read_depth_png16(path: str) -> np.ndarray:
    from PIL import Image

    return np.asarray(Image.open(path)).astype(np.uint16)


class EvaluationRenderer:
    """``dataset`` provides ``camera_intrinsics`` (dict with fx, fy, cx, cy, height, width, depth_scale),
    ``camera_extrinsics`` ([N,4,4] ground-truth c2w, standard axes), ``evaluation_frames``, ``num_frames`` and
    ``frames_depth(mode=, keyframes=)``; ``nerf`` is a ``NeRFRenderer``; ``keyframes`` the dataset indices of the
    training keyframes (renderer.py:33-63)."""

    def __init__(self, dataset, nerf, keyframes, dir_prediction: str) -> None:
        self.dataset = dataset
        self.nerf = nerf
        self.keyframes = list(keyframes)
        self.dir_prediction = dir_prediction
        self._calculate_pred2gt_transformation()

    def _process_mode(self, mode: str) -> tuple:
        if mode == "evaluation_frames":
            return "evaluation_frames", list(self.dataset.evaluation_frames)
        if mode == "keyframes":
            return "keyframes", list(self.keyframes)
        if mode == "all":
            return "all_frames", list(range(self.dataset.num_frames))  # (the reference's len(int) would raise)
        raise NotImplementedError(mode)

    def _calculate_pred2gt_transformation(self) -> None:
        depths_gt = self.dataset.frames_depth(mode="keyframes", keyframes=self.keyframes)
        depths_pred = [self.nerf.render_frame_depth_from_training_frame(
            camera_intrinsics=self.dataset.camera_intrinsics, frame_index=i) for i in range(len(depths_gt))]
        self.pred2gt_transformation = estimate_pred2gt(
            depths_gt, depths_pred, np.asarray(self.dataset.camera_extrinsics[0], dtype=np.float64),
            self.nerf.get_camera_extrinsics(frame_index=0))

    def _render_frame(self, camera_extrinsics: np.ndarray, file_color: str, file_depth: str) -> None:
        color, depth = self.nerf.render_frame(camera_intrinsics=self.dataset.camera_intrinsics,
                                              camera_extrinsics=camera_extrinsics)
        units = depth * self.pred2gt_transformation["scale_pred2gt"] * self.dataset.camera_intrinsics["depth_scale"]
        write_color_jpeg(file_color, color)
        write_depth_png16(file_depth, units)

    def render_frames(self, mode: str = "evaluation_frames") -> list:
        folder, indices = self._process_mode(mode)
        os.makedirs(f"{self.dir_prediction}/{folder}/color", exist_ok=True)
        os.makedirs(f"{self.dir_prediction}/{folder}/depth", exist_ok=True)
        gt = np.stack([np.asarray(self.dataset.camera_extrinsics[i], dtype=np.float64) for i in indices])
        for index, pose in zip(indices, transform_camera_extrinsics_gt2pred(gt, self.pred2gt_transformation)):
            self._render_frame(pose, f"{self.dir_prediction}/{folder}/color/{index:06d}.jpg",
                               f"{self.dir_prediction}/{folder}/depth/{index:06d}.png")
        return indices

    def export_keyframe_poses(self) -> np.ndarray:
        """matrices/matrices_origin2frame_keyframes_mapping.json: training poses with the translation in
        ground-truth metres (renderer.py:225-236)."""
        poses = np.stack([self.nerf.get_camera_extrinsics(frame_index=i) for i in range(len(self.keyframes))])
        poses[:, :3, 3] *= self.pred2gt_transformation["scale_pred2gt"]
        os.makedirs(f"{self.dir_prediction}/matrices", exist_ok=True)
        with open(f"{self.dir_prediction}/matrices/matrices_origin2frame_keyframes_mapping.json", "w") as file:
            json.dump(poses.tolist(), file)
        return poses


class Evaluator2D:
    """Per-frame depth + colour metrics of the frames ``EvaluationRenderer.render_frames`` wrote
    (evaluator.py:88-146): ``metrics_2d_<folder>.csv`` (one row per frame) and ``.json`` (column means)."""

    def __init__(self, dataset, keyframes, dir_prediction: str, dir_result: str, lpips_loss=None) -> None:
        self.dataset = dataset
        self.keyframes = list(keyframes)
        self.dir_prediction = dir_prediction
        self.dir_result = dir_result
        self.lpips_loss = lpips_loss

    def calculate_metrics_2d(self, mode: str = "evaluation_frames") -> dict:
        import pandas as pd

        folder = {"evaluation_frames": "evaluation_frames", "keyframes": "keyframes", "all": "all_frames"}[mode]
        colors_gt = self.dataset.frames_color(mode=mode, keyframes=self.keyframes)
        depths_gt = self.dataset.frames_depth(mode=mode, keyframes=self.keyframes)
        color_dir, depth_dir = f"{self.dir_prediction}/{folder}/color", f"{self.dir_prediction}/{folder}/depth"
        colors_pred = [read_color(os.path.join(color_dir, f)) for f in sorted(os.listdir(color_dir)) if f.endswith(".jpg")]
        depths_pred = [read_depth_png16(os.path.join(depth_dir, f)) / self.dataset.camera_intrinsics["depth_scale"]
                       for f in sorted(os.listdir(depth_dir)) if f.endswith(".png")]
        rows = []
        for d_gt, d_pred, c_gt, c_pred in zip(depths_gt, depths_pred, colors_gt, colors_pred):
            row = dict(calculate_depth_metrics_2d(frame_depth_gt=d_gt, frame_depth_pred=d_pred))
            row.update(calculate_color_metrics_2d(frame_color_gt=c_gt, frame_color_pred=c_pred,
                                                  lpips_loss=self.lpips_loss))
            rows.append(row)
        table = pd.DataFrame(rows)
        os.makedirs(self.dir_result, exist_ok=True)
        table.to_csv(f"{self.dir_result}/metrics_2d_{folder}.csv", index=False)
        means = table.mean().to_dict()
        with open(f"{self.dir_result}/metrics_2d_{folder}.json", "w") as file:
            json.dump(means, file)
        return means
