"""End-to-end 2-D evaluation of a natively trained model: train through the
Nerfstudio mapper interface, render evaluation frames to disk through NerfstudioRenderer, score them with the
reference's metrics."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_evaluation_loop_on_trained_model(device, tmp_path):
    from nerf_vo_amd.evaluation import EvaluationRenderer, Evaluator2D
    from nerf_vo_amd.mapping.dataset import opencv_to_opengl
    from nerf_vo_amd.mapping.nerfstudio_mapper import Nerfstudio
    from nerf_vo_amd.mapping.renderer import NerfstudioRenderer
    from nerf_vo_amd.synthetic import SyntheticEvaluationDataset

    H, W, iterations = 120, 160, 1500
    ds = SyntheticEvaluationDataset(num_frames=96, height=H, width=W)
    keyframes = list(range(0, 96, 2))
    args = argparse.Namespace(experiment="eval", dir_prediction=str(tmp_path / "pred"),
                              mapping_snapshot_iterations=iterations, mapping_iterations=iterations,
                              num_keyframes=len(keyframes), frame_height=H, frame_width=W, enhancement_module="depth")
    mapper = Nerfstudio(args, device=device)
    ci = ds.camera_intrinsics
    colors = torch.stack([torch.from_numpy(c) for c in ds.frames_color("keyframes", keyframes)]).permute(0, 3, 1, 2)
    depths = torch.stack([torch.from_numpy(d) for d in ds.frames_depth("keyframes", keyframes)])[:, None]
    poses = torch.from_numpy(ds.camera_extrinsics[keyframes]).float()
    mapper(input={
        "keyframe_indices": torch.arange(len(keyframes)),
        "camera_intrinsics": torch.tensor([ci["fx"], ci["fy"], ci["cx"], ci["cy"]]).repeat(len(keyframes), 1).to(device),
        "camera_extrinsics": opencv_to_opengl(poses.to(device)),
        "frames_color": (colors.float() / 255.0).to(device), "frames_depth": depths.float().to(device),
        "last_frame": True})
    while mapper.step < iterations:
        mapper(input=None)
    mapper(input=None)
    assert mapper.is_shut_down

    nerf = NerfstudioRenderer(mapping_model=mapper)
    renderer = EvaluationRenderer(dataset=ds, nerf=nerf, keyframes=keyframes, dir_prediction=str(tmp_path / "pred"))
    # the mapper's world is the ground truth's up to the frame-0 normalisation: metric scale is preserved
    assert renderer.pred2gt_transformation["scale_pred2gt"] == pytest.approx(1.0, abs=0.03)
    indices = renderer.render_frames(mode="evaluation_frames")
    assert len(os.listdir(tmp_path / "pred" / "evaluation_frames" / "depth")) == len(indices)
    renderer.export_keyframe_poses()
    metrics = Evaluator2D(ds, keyframes, str(tmp_path / "pred"), str(tmp_path / "res")).calculate_metrics_2d()
    print(metrics)
    assert np.isfinite(list(metrics.values())).all()
    assert metrics["delta1"] > 0.97 and metrics["absolute_difference"] < 0.12
    assert metrics["psnr"] > 27.0 and metrics["mssim"] > 0.80  # reference (uint8-wrapping) PSNR definition
