"""CPU tests of the host logic around the hot path: keyframe ingest semantics of the DynamicDataset
mirror (/root/reference/nerf_vo/mapping/nerfstudio_utils.py:157-228), the mapper cadence helpers and
the synthetic sequence generator."""
import os

import numpy as np
import pytest
import torch


def _dataset(**kw):
    from nerf_vo_amd.mapping.dataset import DynamicDataset

    args = dict(num_frames=8, frame_height=6, frame_width=8, device=torch.device("cpu"), use_normals=True)
    args.update(kw)
    return DynamicDataset(**args)


def _pose(seed):
    g = torch.Generator().manual_seed(seed)
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    if torch.det(q) < 0:
        q[:, 0] *= -1
    m = torch.eye(4)
    m[:3, :3] = q
    m[:3, 3] = torch.randn(3, generator=g)
    return m


def _frames(n, seed, h=6, w=8):
    g = torch.Generator().manual_seed(seed)
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, h, w, generator=g), dim=1)
    return torch.rand(n, 3, h, w, generator=g), torch.rand(n, 1, h, w, generator=g) * 5, nrm


def test_dense_ingest_indexes_by_keyframe_and_normalises_world():
    ds = _dataset()
    poses = torch.stack([_pose(i) for i in range(3)])
    color, depth, normal = _frames(3, 0)
    ds.update({"keyframe_indices": torch.tensor([0, 1, 2]), "camera_intrinsics": torch.rand(3, 4),
               "camera_extrinsics": poses, "frames_color": color, "frames_depth": depth, "frames_normal": normal})
    assert ds.num_active_frames == 3 and len(ds) == 3
    # reference semantics (nerfstudio_utils.py:189-199): N = solve(E_0, M) = inv(E_0) @ M is computed
    # once from the first pose and EVERY pose becomes N @ E_k (so the first one is inv(E_0) M E_0)
    M = torch.tensor([[1.0, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])
    N = torch.linalg.inv(poses[0]) @ M
    assert torch.allclose(ds.normalization_matrix, N, atol=1e-5)
    assert torch.allclose(ds.camera_extrinsics[0], N @ poses[0], atol=1e-5)
    assert torch.allclose(ds.camera_extrinsics[2], N @ poses[2], atol=1e-5)
    # NCHW -> NHWC
    assert torch.equal(ds.frames_color[1], color[1].permute(1, 2, 0))
    assert torch.equal(ds.frames_depth[2], depth[2].permute(1, 2, 0))
    # cameras hold VIEWS of the buffers: later in-place updates are visible (aliasing contract)
    before = ds.cameras.camera_to_worlds[1].clone()
    ds.camera_extrinsics[1, :3, 3] += 1.0
    assert not torch.equal(ds.cameras.camera_to_worlds[1], before)


def test_sparse_ingest_appends_frames_and_refreshes_window():
    ds = _dataset()
    color, depth, normal = _frames(2, 1)
    poses = torch.stack([_pose(i) for i in range(2)])
    ds.update({"keyframe_indices": torch.tensor([0, 1]), "camera_intrinsics": torch.rand(2, 4),
               "camera_extrinsics": poses, "frames_color": color, "frames_depth": depth, "frames_normal": normal})
    # tracker-style packet: ONE new frame, poses + depths refreshed for the window [0,1,2]
    color2, _, normal2 = _frames(1, 2)
    _, depth_win, _ = _frames(3, 3)
    poses_win = torch.stack([_pose(10 + i) for i in range(3)])
    old_color0 = ds.frames_color[0].clone()
    ds.update({"keyframe_indices": torch.tensor([0, 1, 2]), "camera_intrinsics": torch.rand(1, 4),
               "camera_extrinsics": poses_win, "frames_color": color2, "frames_depth": depth_win,
               "frames_normal": normal2})
    assert ds.num_active_frames == 3
    assert torch.equal(ds.frames_color[0], old_color0)                      # old colours untouched
    assert torch.equal(ds.frames_color[2], color2[0].permute(1, 2, 0))     # new frame appended at slot 2
    assert torch.equal(ds.frames_depth[0], depth_win[0].permute(1, 2, 0))  # depth refreshed for the window
    N = ds.normalization_matrix
    assert torch.allclose(ds.camera_extrinsics[1], N @ poses_win[1], atol=1e-5)


def test_world_normals_are_cached_and_match_per_step_solve():
    ds = _dataset()
    poses = torch.stack([_pose(i) for i in range(2)])
    color, depth, normal = _frames(2, 4)
    ds.update({"keyframe_indices": torch.tensor([0, 1]), "camera_intrinsics": torch.rand(2, 4),
               "camera_extrinsics": poses, "frames_color": color, "frames_depth": depth, "frames_normal": normal})
    got = ds.get_dataset()["normal_image"]
    # the reference recomputes (R^-1 n + 1) / 2 on every get_dataset() call (nerfstudio_utils.py:145-153)
    rot = ds.camera_extrinsics[:2, :3, :3]
    n = ds.frames_normal[:2].permute(0, 3, 1, 2).reshape(2, 3, -1)
    ref = (torch.linalg.solve(rot, n).reshape(2, 3, 6, 8).permute(0, 2, 3, 1) + 1) / 2
    assert torch.allclose(got, ref, atol=1e-5)
    assert ds.get_frame(1)["normal_image"].shape == (6, 8, 3)


def test_keyframe_index_beyond_buffer_is_rejected():
    ds = _dataset(num_frames=2)
    color, depth, normal = _frames(1, 5)
    with pytest.raises(AssertionError):
        ds.update({"keyframe_indices": torch.tensor([2]), "camera_intrinsics": torch.rand(1, 4),
                   "camera_extrinsics": _pose(0)[None], "frames_color": color, "frames_depth": depth,
                   "frames_normal": normal})


def test_save_and_reload_dataset(tmp_path):
    ds = _dataset(use_normals=False)
    color, depth, _ = _frames(2, 6)
    ds.update({"keyframe_indices": torch.tensor([0, 1]), "camera_intrinsics": torch.rand(2, 4),
               "camera_extrinsics": torch.stack([_pose(0), _pose(1)]), "frames_color": color,
               "frames_depth": depth})
    ds.save_dataset(str(tmp_path))
    ds2 = _dataset(use_normals=False, dir_prediction=str(tmp_path))
    assert ds2.num_active_frames == 2
    assert torch.equal(ds2.frames_color[:2], ds.frames_color[:2])
    assert torch.equal(ds2.camera_extrinsics[:2], ds.camera_extrinsics[:2])


def test_pixel_sampler_range_and_rank_streams():
    from nerf_vo_amd.mapping.dataset import DynamicDataManager, DynamicDataManagerConfig

    cfg = DynamicDataManagerConfig(train_num_rays_per_batch=512, num_frames=4, frame_height=6, frame_width=8,
                                   use_normals=False)
    dms = [DynamicDataManager(cfg, device=torch.device("cpu"), world_size=2, local_rank=r) for r in (0, 1)]
    for dm in dms:
        dm.train_dataset.num_active_frames = 3
    idx0, _ = dms[0].next_train(0)
    idx1, _ = dms[1].next_train(0)
    assert idx0.shape == (512, 3) and idx0.dtype == torch.int64
    assert int(idx0[:, 0].max()) < 3 and int(idx0[:, 1].max()) < 6 and int(idx0[:, 2].max()) < 8
    assert not torch.equal(idx0, idx1), "ranks must draw different rays (weak scaling)"


def test_step_check_and_schedules():
    from nerf_vo_amd.mapping.nerfstudio_mapper import step_check
    from oracle import rays as Rr

    assert not step_check(0, 10) and step_check(0, 10, run_at_zero=True) and step_check(20, 10)
    assert not step_check(5, 0)
    assert Rr.proposal_anneal(0) == 0.0 and Rr.proposal_anneal(1000) == pytest.approx(1.0)
    assert Rr.proposal_update_due(3, 0) and not Rr.proposal_update_due(100, 1) and Rr.proposal_update_due(100, 2)
    assert not Rr.proposal_update_due(6000, 5) and Rr.proposal_update_due(6000, 6)


def test_synthetic_sequence_shapes_and_geometry():
    from nerf_vo_amd.synthetic import make_sequence, replica_intrinsics

    seq = make_sequence(4, 30, 40)
    assert seq["frames_color"].shape == (4, 3, 30, 40) and seq["frames_depth"].shape == (4, 1, 30, 40)
    assert float(seq["frames_color"].min()) >= 0 and float(seq["frames_color"].max()) <= 1
    assert float(seq["frames_depth"].min()) > 0 and float(seq["frames_depth"].max()) <= 5
    n = seq["frames_normal"]
    assert torch.allclose(n.norm(dim=1), torch.ones(4, 30, 40), atol=1e-5)
    # 640x480 -> fx 320.0, fy 423.529, cx 319.733, cy 239.647
    fx, fy, cx, cy = replica_intrinsics(480, 640)
    assert (fx, round(fy, 3), round(cx, 3), round(cy, 3)) == (320.0, 423.529, 319.733, 239.647)
    rot = seq["camera_extrinsics"][:, :3, :3]
    assert torch.allclose(rot @ rot.transpose(1, 2), torch.eye(3).expand(4, 3, 3), atol=1e-5)


# ---- host logic pinned by the reference itself (tests/golden/make_golden_host.py imports /root/reference) ----
def _host_golden():
    import json
    import os

    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_golden.json")))


def test_mapping_module_cadence_matches_reference():
    """MappingModule.step (ref: nerf_vo/mapping/mapping_module.py:35-55): which ticks train, which are skipped."""
    from nerf_vo_amd.mapping.mapping_module import MappingModule

    class Fake:
        def __init__(self, stop):
            self.calls, self.is_shut_down, self.stop = [], False, stop

        def __call__(self, input):
            self.calls.append(input is not None)
            if len(self.calls) >= self.stop:
                self.is_shut_down = True

    for case in _host_golden()["cadence"]:
        mod = MappingModule(Fake(case["shut_down_after"]), case["mapping_iterations"], case["num_keyframes"])
        rows = []
        for item in case["schedule"]:
            inp = None if item == 0 else {"last_frame": item == 2}
            n = len(mod.method.calls)
            _, skip = mod.step(inp)
            rows.append([item, int(len(mod.method.calls) > n), int(skip), int(mod.shutdown)])
        assert rows == case["rows"]


def test_replica_intrinsics_match_reference_scaling():
    """synthetic.replica_intrinsics vs scale_camera_intrinsics(datasets/replica.json) of the reference
    (ref: nerf_vo/data/data_utils.py:24-34)."""
    from nerf_vo_amd.synthetic import replica_intrinsics

    for row in _host_golden()["intrinsics"]:
        fx, fy, cx, cy = replica_intrinsics(row["height"], row["width"])
        assert (fx, fy, cx, cy) == (row["fx"], row["fy"], row["cx"], row["cy"])


def test_psnr_reference_definition_matches_reference_function():
    """calculate_psnr_reference (product) and oracle.rays.psnr_reference vs the reference's own calculate_psnr
    evaluated per channel (ref: evaluation/evaluation_utils.py:289-318), uint8 wrap-around included."""
    import numpy as np

    from nerf_vo_amd.mapping.renderer import calculate_psnr_reference
    from oracle.rays import psnr_reference

    for row in _host_golden()["psnr"]:
        a, b = np.asarray(row["a"], dtype=np.uint8), np.asarray(row["b"], dtype=np.uint8)
        assert calculate_psnr_reference(a, b) == pytest.approx(row["color"], rel=1e-12)
        assert psnr_reference(a, b) == pytest.approx(row["color"], rel=1e-12)


def test_runtime_log_csv_matches_reference_format(tmp_path):
    """PerformanceTracker tuples -> RuntimeLog -> runtime_<process>.csv, byte-identical to what the reference's
    ``pd.DataFrame(rows).to_csv(path, index=False)`` writes for the same rows (logging_module.py:21-25,46-57)."""
    import queue

    import pandas as pd

    from nerf_vo_amd.runtime_log import PerformanceTracker, RuntimeLog

    q = queue.Queue()
    expected = {"mapping": [], "tracking": []}
    for step in range(5):
        for name in ("mapping", "tracking"):
            with PerformanceTracker(process_name=name, logging_queue=q, step=step) as tracker:
                sum(range(1000 * (step + 1)))
            tracker.submit()
            assert tracker.runtime >= 0.0
            expected[name].append({"step": step, "runtime": tracker.runtime})
    q.put(("mapping", "loss", 3, 0.5))  # non-runtime fields are ignored (wandb is commented out upstream)
    q.put(("logging", "shutdown", 0, 0.0))
    log = RuntimeLog()
    log.step(log.drain(q))
    assert log.shutdown and q.empty()
    written = log.shut_down(str(tmp_path))
    assert sorted(os.path.basename(p) for p in written) == ["runtime_mapping.csv", "runtime_tracking.csv"]
    for name, rows in expected.items():
        pd.DataFrame(rows).to_csv(tmp_path / f"ref_{name}.csv", index=False)
        assert (tmp_path / f"runtime_{name}.csv").read_text() == (tmp_path / f"ref_{name}.csv").read_text()
    PerformanceTracker("x", None, 0).submit()  # no queue: a no-op, like the reference


def test_pyngp_facade_surface_and_loud_failure_without_gpu():
    """The testbed facade exposes the names the reference touches (instant_ngp.py:33-48, nerf_renderer.py:263-316) and
    refuses to run without an MI355X instead of falling back to anything."""
    from nerf_vo_amd import pyngp

    assert pyngp.TestbedMode.Nerf and pyngp.LossType.L2 and pyngp.Shade != pyngp.Depth
    box = pyngp.BoundingBox(np.array([-np.inf] * 3), np.array([np.inf] * 3))
    assert box.min.shape == (3,) and np.isinf(box.max).all()
    for name in ("create_empty_nerf_dataset", "reload_network_from_file", "frame", "save_snapshot", "load_snapshot",
                 "set_nerf_camera_matrix", "render", "compute_and_save_marching_cubes_mesh"):
        assert callable(getattr(pyngp.Testbed, name))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            pyngp.Testbed(pyngp.TestbedMode.Nerf, 0)
