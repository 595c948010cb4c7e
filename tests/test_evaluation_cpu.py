"""2-D evaluation loop against golden vectors the REFERENCE produced
(tests/golden/make_golden_evaluation.py executes its definitions), plus the file round trip."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "evaluation_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_depth_metrics_match_reference(gold):
    from nerf_vo_amd.evaluation import calculate_depth_metrics_2d

    names = [str(n) for n in gold["depth_metric_names"]]
    for i in range(3):
        gt, pred = gold[f"depth{i}_gt"], gold[f"depth{i}_pred"]
        pred_before = pred.copy()
        for with_scale in (True, False):
            m = calculate_depth_metrics_2d(gt, pred, with_scale=with_scale)
            assert sorted(m) == names
            # float64 numpy on both sides; only the association of the scale multiply differs: rtol 1e-12
            np.testing.assert_allclose([m[k] for k in names], gold[f"depth{i}_metrics_scale{int(with_scale)}"], rtol=1e-12)
        assert np.array_equal(pred, pred_before), "inputs must not be modified"


def test_mssim_matches_reference(gold):
    from nerf_vo_amd.evaluation import calculate_mssim

    for i in range(3):
        v = calculate_mssim(torch.tensor(gold[f"ssim{i}_a"]), torch.tensor(gold[f"ssim{i}_b"]))
        # separable vs 11x11 window in float32: tolerance 2e-6 absolute on an SSIM in [-1, 1]
        assert abs(v - float(gold[f"ssim{i}_value"])) < 2e-6, (i, v, float(gold[f"ssim{i}_value"]))
    assert calculate_mssim(torch.tensor(gold["ssim2_a"]), torch.tensor(gold["ssim2_a"])) == pytest.approx(1.0, abs=1e-6)


def test_pred2gt_alignment_matches_reference(gold):
    from nerf_vo_amd.evaluation import estimate_pred2gt, transform_camera_extrinsics_gt2pred, transform_matrices_pred2gt

    tf = estimate_pred2gt(list(gold["align_depths_gt"]), list(gold["align_depths_pred"]), gold["align_extr_gt"][0],
                          gold["align_extr_pred0"])
    assert float(tf["scale_pred2gt"]) == float(gold["align_scale"])  # same masked means, same median: bit-exact
    np.testing.assert_allclose(tf["matrix_pred2gt"], gold["align_matrix"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(tf["matrix_pred2gt_scaled"], gold["align_matrix_scaled"], rtol=0, atol=1e-14)
    g2p = transform_camera_extrinsics_gt2pred(gold["align_extr_gt"], tf)
    np.testing.assert_allclose(g2p, gold["align_gt2pred"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(transform_matrices_pred2gt(g2p, tf), gold["align_pred2gt"], rtol=0, atol=1e-13)
    # round trip: gt -> pred -> gt is the identity on the poses
    np.testing.assert_allclose(transform_matrices_pred2gt(g2p, tf), gold["align_extr_gt"], rtol=0, atol=1e-12)


class _FakeNerf:
    """Renders the analytic room through the renderer interface (no GPU): what a perfect model would return,
    in a world that is rotated, shifted and shrunk relative to the ground truth."""

    def __init__(self, dataset, keyframes, world_from_gt, shrink):
        self.dataset, self.keyframes, self.world_from_gt, self.shrink = dataset, keyframes, world_from_gt, shrink

    def _to_world(self, pose_gt):
        p = self.world_from_gt @ pose_gt
        p[:3, 3] *= self.shrink
        return p

    def get_camera_extrinsics(self, frame_index):
        return self._to_world(self.dataset.camera_extrinsics[self.keyframes[frame_index]])

    def render_frame(self, camera_intrinsics, camera_extrinsics):
        pose = camera_extrinsics.copy()
        pose[:3, 3] /= self.shrink
        pose_gt = np.linalg.inv(self.world_from_gt) @ pose
        color, depth = self.dataset.render(pose_gt)
        return color, depth * self.shrink

    def render_frame_depth_from_training_frame(self, camera_intrinsics, frame_index):
        return self.render_frame(camera_intrinsics, self.get_camera_extrinsics(frame_index))[1]


def test_evaluation_loop_round_trip(tmp_path):
    """EvaluationRenderer + Evaluator2D on a perfect 'model' living in a similarity-transformed world: the
    alignment must recover the scale, rendered files must land where the reference puts them and the metrics must
    be those of a JPEG / 16-bit-PNG round trip of the ground truth."""
    from nerf_vo_amd.evaluation import EvaluationRenderer, Evaluator2D
    from nerf_vo_amd.synthetic import SyntheticEvaluationDataset

    ds = SyntheticEvaluationDataset(num_frames=24, height=60, width=80)
    keyframes = list(range(0, 24, 4))
    rot = np.eye(4)
    a = 0.7
    rot[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
    rot[:3, 3] = [0.3, -0.2, 0.1]
    nerf = _FakeNerf(ds, keyframes, rot, shrink=0.25)
    renderer = EvaluationRenderer(dataset=ds, nerf=nerf, keyframes=keyframes, dir_prediction=str(tmp_path / "pred"))
    assert renderer.pred2gt_transformation["scale_pred2gt"] == pytest.approx(4.0, rel=1e-6)
    indices = renderer.render_frames(mode="evaluation_frames")
    assert indices == list(ds.evaluation_frames)
    assert sorted(os.listdir(tmp_path / "pred" / "evaluation_frames" / "color")) == [f"{i:06d}.jpg" for i in indices]
    poses = renderer.export_keyframe_poses()
    # exported keyframe translations are in ground-truth metres
    np.testing.assert_allclose(np.linalg.norm(poses[1, :3, 3] - poses[0, :3, 3]),
                               np.linalg.norm(ds.camera_extrinsics[keyframes[1], :3, 3] - ds.camera_extrinsics[keyframes[0], :3, 3]),
                               rtol=1e-6)
    metrics = Evaluator2D(ds, keyframes, str(tmp_path / "pred"), str(tmp_path / "res")).calculate_metrics_2d("evaluation_frames")
    assert metrics["absolute_difference"] < 2e-3 and metrics["delta1"] == 1.0  # 1/depth_scale quantisation only
    assert metrics["psnr"] > 33.0 and metrics["mssim"] > 0.97  # JPEG q95 of a smooth image
    assert os.path.exists(tmp_path / "res" / "metrics_2d_evaluation_frames.csv")
    assert os.path.exists(tmp_path / "res" / "metrics_2d_evaluation_frames.json")
