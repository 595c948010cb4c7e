#!/usr/bin/env python3
"""The reference's PUBLISHED evaluation protocol on a natively trained model.

Train through the Nerfstudio mapper interface (incremental keyframe ingest), then evaluate exactly as
/root/reference/run.py:57-79 does: ``Renderer`` (evaluation/renderer.py) aligns prediction and ground truth through
FRAME 0's pose and the MEDIAN per-keyframe depth scale (:79-111), keyframe views are rendered at the model's own
(optimised) training poses (evaluation/nerf_renderer.py:125-130), evaluation frames at ground-truth poses carried into
the model's world (renderer.py:276-287), everything goes through JPEG / 16-bit PNG files (:118-124), and ``Evaluator``
scores the files (evaluator.py:88-146; PSNR = the reference's uint8-wrapping definition).  Here: nerf_vo_amd.evaluation's
EvaluationRenderer / Evaluator2D, whose numerics are pinned by the reference's own functions (tests/golden/
make_golden_evaluation.py).

``pose_noise = (sigma_rot [rad], sigma_trans)``: the ingested keyframe poses are perturbed the way a tracker's would be
(frame 0 exact: it anchors the world) -- BASELINE configs[2], where the SE3 camera optimiser has something to recover.
Reports, next to the 2-D metrics, the pose error of the exported keyframe trajectory after the protocol's frame-0
alignment (translation RMSE, mean rotation angle) with and without the optimiser's corrections.

    python tools/eval_protocol.py --keyframes 48 --height 120 --width 160 --iterations 1500 --camera-optimizer-mode SE3 \\
        --pose-noise 5e-3 5e-3
"""
import argparse
import json
import math
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def _so3_exp(w: torch.Tensor) -> torch.Tensor:
    th = w.norm()
    k = torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=w.dtype)
    if th < 1e-12:
        return torch.eye(3, dtype=w.dtype) + k
    return torch.eye(3, dtype=w.dtype) + torch.sin(th) / th * k + (1 - torch.cos(th)) / th ** 2 * (k @ k)


def _pose_errors(pred: np.ndarray, gt: np.ndarray) -> dict:
    dt = pred[:, :3, 3] - gt[:, :3, 3]
    rr = np.einsum("nij,nkj->nik", pred[:, :3, :3], gt[:, :3, :3])  # R_pred R_gt^T
    ang = np.arccos(np.clip((np.trace(rr, axis1=1, axis2=2) - 1) / 2, -1, 1))
    return {"translation_rmse": float(np.sqrt((dt ** 2).sum(1).mean())), "rotation_mean_rad": float(ang.mean())}


def _gauge_split(pred: np.ndarray, gt: np.ndarray) -> dict:
    """The pose error split into what a common rigid motion of ALL cameras explains (a gauge: the field absorbs it, and
    the protocol's alignment pins it on frame 0 alone) and what is left.  D_i = pred_i gt_i^-1 is camera i's error as a
    world-frame motion; G = their mean (rotation vectors and translations averaged: the errors are ~1e-3); the residual of
    camera i is G^-1 D_i."""
    n = pred.shape[0]
    P4, G4 = np.tile(np.eye(4), (n, 1, 1)), np.tile(np.eye(4), (n, 1, 1))
    P4[:, :3, :4], G4[:, :3, :4] = pred[:, :3, :4], gt[:, :3, :4]
    D = P4 @ np.linalg.inv(G4)

    def log_so3(R):
        ang = np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))
        w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
        return w * (0.5 if ang < 1e-8 else ang / (2 * np.sin(ang)))

    def exp_so3(w):
        th = np.linalg.norm(w)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        return np.eye(3) + K if th < 1e-8 else np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)

    w = np.stack([log_so3(D[i, :3, :3]) for i in range(n)])
    Gm = np.eye(4)
    Gm[:3, :3], Gm[:3, 3] = exp_so3(w.mean(0)), D[:, :3, 3].mean(0)
    Res = np.linalg.inv(Gm)[None] @ D
    res_ang = np.array([np.linalg.norm(log_so3(Res[i, :3, :3])) for i in range(n)])
    return {"mean_rigid_rotation_rad": float(np.linalg.norm(w.mean(0))), "mean_rigid_translation": float(np.linalg.norm(Gm[:3, 3])),
            "residual_rotation_mean_rad": float(res_ang.mean()),
            "residual_translation_rmse": float(np.sqrt((Res[:, :3, 3] ** 2).sum(1).mean())),
            "raw_rotation_mean_rad": float(np.linalg.norm(w, axis=1).mean())}


def _log(mapper):
    every = int(os.environ.get("NVO_PROTO_LOG", "0"))
    lo, hi = int(os.environ.get("NVO_PROTO_LOG_FROM", "0")), int(os.environ.get("NVO_PROTO_LOG_TO", "1000000000"))
    if every and mapper.step % every == 0 and lo <= mapper.step <= hi:
        eng = mapper.trainer.pipeline.model.engine
        torch.cuda.synchronize()
        if os.environ.get("NVO_PROTO_LOG_GRADS"):
            bad = {n: int((~torch.isfinite(eng.grads[o:o + sz])).sum()) for n, (o, sz, _) in eng.segments.items()}
            ws = eng._workspace(eng.cfg.num_rays, True)
            extra = {k: int((~torch.isfinite(ws[k].float())).sum()) for k in ("drgb", "dout2", "dout0", "dout1", "out2", "rgb", "weights2", "x2")}
            mx = {n: float(eng.params[o:o + sz].abs().max()) for n, (o, sz, _) in eng.segments.items()}
            gmax = {n: float((eng.grads[o:o + sz] / eng.current_loss_scale()).abs().max()) for n, (o, sz, _) in eng.segments.items() if sz}
            print(f"   max|grad| {gmax}", file=sys.stderr, flush=True)
            pre = ws["out2"][:, 0].float()
            d2 = ws["dout2"].float()
            rows = (~torch.isfinite(d2)).any(dim=1).nonzero().flatten()[:4]
            detail = [(int(r), [float(v) for v in d2[r][:3]], float(pre[r]), float(ws["weights2"][r]),
                       [float(v) for v in ws["tbins2"].view(-1, 49)[r // 48, (r % 48):(r % 48) + 2]]) for r in rows.tolist()]
            print(f"   nonfinite grads {bad} ws {extra} max pre {float(pre.max()):.2f} n(pre>14) {int((pre > 14).sum())} bad rows {detail}", file=sys.stderr, flush=True)
        print(f"it {mapper.step} scale {eng.current_loss_scale()} tracker {int(eng.dev_growth_tracker.item())} flags "
              f"{eng.skip_flag.tolist()} applied {eng.opt_steps} finite {bool(torch.isfinite(eng.params).all())} "
              f"half_finite {bool(torch.isfinite(eng.working_copy_float()).all())} losses {eng.loss_dict()}", file=sys.stderr, flush=True)


def run(keyframes=48, height=120, width=160, iterations=1500, frame_stride=2, eval_frames=8, chunk=8, device="cuda:0",
        camera_optimizer_mode=None, pose_noise=None, deterministic=False, seed=42, dynamic_loss_scale=None, out_dir=None,
        quiet=True, keyframe_views=True, method="nerfstudio", scene_scale=None):
    """``method``: 'nerfstudio' (the default mapper) or 'instant-ngp' (the occupancy-grid back-end through the pyngp facade,
    /root/reference/nerf_vo/mapping/instant_ngp.py + evaluation/nerf_renderer.py:221-320; the room is shrunk by
    ``scene_scale``, default 0.5, so that it lies inside that back-end's scene box: it takes poses as they come)."""
    entry.build()
    from nerf_vo_amd.evaluation import (EvaluationRenderer, Evaluator2D, read_color, transform_matrices_pred2gt)
    from nerf_vo_amd.mapping.dataset import opencv_to_opengl
    from nerf_vo_amd.mapping.nerfstudio_mapper import Nerfstudio
    from nerf_vo_amd.mapping.renderer import NerfstudioRenderer, calculate_psnr_float
    from nerf_vo_amd.synthetic import SyntheticEvaluationDataset

    torch.manual_seed(int(seed))
    dev = torch.device(device)
    own_dir = out_dir is None  # (a directory of our own making is removed again: 1.1 GB of frames and snapshots per run)
    out_dir = out_dir or tempfile.mkdtemp(prefix="nvo_eval_")
    n_frames = keyframes * frame_stride  # dataset frames; every frame_stride-th one is a keyframe (configs: frame_stride 2)
    ngp = method == "instant-ngp"
    if ngp and abs(width / height - 1200.0 / 680.0) > 0.01:
        # (the testbed's free camera has ONE focal length, evaluation/nerf_renderer.py:152: Replica's intrinsics have square
        # pixels only at its own aspect -- 160 x 120 renders with fx != fy come out at 13 dB)
        raise SystemExit("--method instant-ngp needs Replica's aspect ratio (e.g. --width 240 --height 136)")
    ds = SyntheticEvaluationDataset(num_frames=n_frames, height=height, width=width, device=dev,
                                    scene_scale=float(scene_scale) if scene_scale else (0.5 if ngp else 1.0))
    kf = list(range(0, n_frames, frame_stride))
    held_out = [i for i in range(n_frames) if i % frame_stride != 0]
    ds.evaluation_frames = [held_out[int(j * len(held_out) / eval_frames)] for j in range(eval_frames)]
    args = argparse.Namespace(experiment="protocol", dir_prediction=out_dir + "/pred", mapping_snapshot_iterations=iterations,
                              mapping_iterations=iterations, num_keyframes=keyframes, frame_height=height, frame_width=width,
                              enhancement_module="depth", deterministic=deterministic, dynamic_loss_scale=dynamic_loss_scale,
                              camera_optimizer_mode=camera_optimizer_mode)
    if ngp:
        from nerf_vo_amd.mapping.instant_ngp_mapper import InstantNGP, InstantNGPRenderer
        mapper = InstantNGP(args, device=dev)
    else:
        mapper = Nerfstudio(args, device=dev)
    ci = ds.camera_intrinsics
    intr = torch.tensor([ci["fx"], ci["fy"], ci["cx"], ci["cy"]])
    poses = torch.from_numpy(ds.camera_extrinsics[kf]).float()
    if pose_noise is not None:  # tracker-like errors on every pose but the first (which anchors the world)
        g = torch.Generator().manual_seed(int(seed) + 1)
        for i in range(1, keyframes):
            poses[i, :3, :3] = _so3_exp(torch.randn(3, generator=g) * pose_noise[0]) @ poses[i, :3, :3]
            poses[i, :3, 3] += torch.randn(3, generator=g) * pose_noise[1]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per_kf = max(int(iterations / keyframes), 1)
    for lo in range(0, keyframes, chunk):
        hi = min(keyframes, lo + chunk)
        frames = [ds.render(ds.camera_extrinsics[i]) for i in kf[lo:hi]]
        color = torch.stack([torch.from_numpy(c) for c, _ in frames]).permute(0, 3, 1, 2).float() / 255.0
        depth = torch.stack([torch.from_numpy(d) for _, d in frames])[:, None].float().clamp(0.0, 5.0)
        gl = opencv_to_opengl(poses[lo:hi].to(dev))
        mapper(input={"keyframe_indices": torch.arange(lo, hi), "camera_intrinsics": intr.repeat(hi - lo, 1).to(dev),
                      "camera_extrinsics": gl, "frames_color": color.to(dev),
                      "frames_depth": depth.to(dev), "last_frame": hi == keyframes})
        for _ in range(per_kf * (hi - lo) - 1):
            if mapper.step < iterations:
                mapper(input=None)
                _log(mapper)
    while mapper.step < iterations:
        mapper(input=None)
        _log(mapper)
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    mapper(input=None)
    assert mapper.is_shut_down

    nerf = InstantNGPRenderer(mapping_model=mapper) if ngp else NerfstudioRenderer(mapping_model=mapper)
    renderer = EvaluationRenderer(dataset=ds, nerf=nerf, keyframes=kf, dir_prediction=args.dir_prediction)
    renderer.render_frames(mode="evaluation_frames")
    exported = renderer.export_keyframe_poses()
    ev = Evaluator2D(ds, kf, args.dir_prediction, out_dir + "/results")
    m_eval = ev.calculate_metrics_2d(mode="evaluation_frames")
    m_kf = {}
    if keyframe_views:  # (every keyframe through JPEG / PNG files and the CPU metrics: seconds per hundred frames)
        renderer.render_frames(mode="keyframes")
        m_kf = ev.calculate_metrics_2d(mode="keyframes")
    # conventional (float-MSE) PSNR of the same JPEG files, beside the reference's definition
    cdir = args.dir_prediction + "/evaluation_frames/color"
    files = sorted(f for f in os.listdir(cdir) if f.endswith(".jpg"))
    gts = ds.frames_color(mode="evaluation_frames", keyframes=kf)
    psnr_float = float(np.mean([calculate_psnr_float(read_color(os.path.join(cdir, f)), g) for f, g in zip(files, gts)]))
    # keyframe trajectory in the ground truth's world through the protocol's frame-0 alignment
    t = renderer.pred2gt_transformation
    gt_kf = np.asarray(ds.camera_extrinsics[kf], dtype=np.float64)
    traj = transform_matrices_pred2gt(np.stack([nerf.get_camera_extrinsics(i) for i in range(keyframes)]), t)
    pose_err = _pose_errors(traj, gt_kf)
    # the same for the poses as INGESTED (no camera-optimiser correction): what the optimiser had to improve on
    ing = poses.double().numpy()
    ing = np.asarray(gt_kf[0])[None] @ np.linalg.inv(ing[0])[None] @ ing  # frame-0 anchored, as the protocol does
    eng = mapper.ngp._engine if ngp else mapper.trainer.pipeline.model.engine
    res = {"method": method, "scene_scale": ds.scene_scale, "keyframes": keyframes, "resolution": [width, height], "iterations": iterations, "train_seconds": train_s,
           "camera_optimizer_mode": camera_optimizer_mode or "SE3", "pose_noise": list(pose_noise) if pose_noise else None,
           "scale_pred2gt": float(t["scale_pred2gt"]),
           "evaluation_frames": {**{k: float(v) for k, v in m_eval.items()}, "psnr_float_mse": psnr_float, "frames": len(files)},
           "keyframe_views": {k: float(v) for k, v in m_kf.items()},
           "pose_error_after_frame0_alignment": pose_err, "pose_error_of_ingested_poses": _pose_errors(ing, gt_kf),
           # the same error split into the cameras' common rigid motion (gauge) and the residual
           "pose_error_gauge_split": _gauge_split(traj, gt_kf),
           "pose_adjustment_rms": float((eng.pose_adjustment if ngp else eng.view("camera_opt.pose_adjustment")).pow(2).mean().sqrt()),
           "deterministic": bool(deterministic), "seed": seed, "exported_poses": int(exported.shape[0])}
    if ngp:
        res.update({"camera_optimizer_mode": "instant-ngp extrinsics", "ms_per_step_incl_ingest": 1e3 * train_s / iterations,
                    "rays_per_batch": eng.rays_per_batch, "applied_steps": eng.applied_steps})
    else:
        res.update({"dynamic_loss_scale": bool(eng.cfg.dynamic_loss_scale), "loss_scale_end": eng.current_loss_scale()})
    if not quiet:
        print(json.dumps(res))
    if own_dir:
        import shutil
        shutil.rmtree(out_dir, ignore_errors=True)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--keyframes", type=int, default=48)
    ap.add_argument("--height", type=int, default=120)
    ap.add_argument("--width", type=int, default=160)
    ap.add_argument("--iterations", type=int, default=1500)
    ap.add_argument("--eval-frames", type=int, default=8)
    ap.add_argument("--camera-optimizer-mode", default=None, help="SE3 (default) | SO3xR3 | off")
    ap.add_argument("--pose-noise", type=float, nargs=2, default=None, metavar=("SIGMA_ROT", "SIGMA_TRANS"))
    ap.add_argument("--deterministic", action="store_true")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--static-loss-scale", action="store_true", help="tcnn's static 128 instead of GradScaler's dynamics")
    ap.add_argument("--no-keyframe-views", action="store_true")
    ap.add_argument("--method", default="nerfstudio", choices=["nerfstudio", "instant-ngp"])
    ap.add_argument("--scene-scale", type=float, default=None)
    a = ap.parse_args()
    run(a.keyframes, a.height, a.width, a.iterations, eval_frames=a.eval_frames, camera_optimizer_mode=a.camera_optimizer_mode,
        pose_noise=a.pose_noise, deterministic=a.deterministic, seed=a.seed, quiet=False,
        dynamic_loss_scale=False if a.static_loss_scale else None, keyframe_views=not a.no_keyframe_views,
        method=a.method, scene_scale=a.scene_scale)
