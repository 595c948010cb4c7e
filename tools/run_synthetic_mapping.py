#!/usr/bin/env python3
"""End-to-end mapping run on the synthetic Replica-shaped sequence, driven through the same interface
the reference's MappingModule uses (mapper(input) / mapper(None), /root/reference/nerf_vo/mapping/
mapping_module.py:36-55), followed by the evaluation render + PSNR of
/root/reference/evaluation/renderer.py:255-263 / evaluation_utils.py:289-318.

Prints one JSON line: training throughput, PSNR (reference-faithful uint8-wrap and float MSE) and
depth L1 on held-out views."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def run(keyframes=48, height=240, width=320, iterations=1500, eval_frames=6, chunk=8, device="cuda:0", quiet=False,
        out_dir=None, enhancement="depth", log_every=0, deterministic=False, dynamic_loss_scale=None, seed=None,
        camera_optimizer_mode=None):
    entry.build()
    if seed is not None:  # (the stateless pixel / jitter sampler follows torch's seed)
        torch.manual_seed(int(seed))
    from nerf_vo_amd.mapping.dataset import opencv_to_opengl
    from nerf_vo_amd.mapping.nerfstudio_mapper import Nerfstudio
    from nerf_vo_amd.mapping.renderer import NerfstudioRenderer, calculate_psnr_float, calculate_psnr_reference
    from nerf_vo_amd.synthetic import make_sequence, orbit_poses_opencv, render_room, replica_intrinsics

    dev = torch.device(device)
    out_dir = out_dir or tempfile.mkdtemp(prefix="nvo_map_")
    args = argparse.Namespace(
        experiment="synthetic", dir_prediction=out_dir, mapping_snapshot_iterations=iterations,
        mapping_iterations=iterations, num_keyframes=keyframes, frame_height=height, frame_width=width,
        enhancement_module=enhancement, deterministic=deterministic, dynamic_loss_scale=dynamic_loss_scale,
        camera_optimizer_mode=camera_optimizer_mode)
    mapper = Nerfstudio(args, device=dev)
    seq = make_sequence(keyframes, height, width, device=dev)
    iters_between = max(int(iterations / keyframes), 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for lo in range(0, keyframes, chunk):
        hi = min(keyframes, lo + chunk)
        mapper(input={
            "keyframe_indices": torch.arange(lo, hi), "camera_intrinsics": seq["camera_intrinsics"][lo:hi],
            "camera_extrinsics": opencv_to_opengl(seq["camera_extrinsics"][lo:hi]),
            "frames_color": seq["frames_color"][lo:hi], "frames_depth": seq["frames_depth"][lo:hi],
            **({"frames_normal": seq["frames_normal"][lo:hi]} if "normal" in enhancement else {}),
            "last_frame": hi == keyframes})
        for _ in range(iters_between * (hi - lo) - 1):
            if mapper.step < iterations:
                mapper(input=None)
                if log_every and mapper.step % log_every == 0:
                    ld = mapper.trainer.pipeline.model.engine.loss_dict()
                    print(f"it {mapper.step:5d} " + " ".join(f"{k}={v:.3e}" for k, v in ld.items()), file=sys.stderr, flush=True)
    while mapper.step < iterations:
        mapper(input=None)
        if log_every and mapper.step % log_every == 0:
            ld = mapper.trainer.pipeline.model.engine.loss_dict()
            print(f"it {mapper.step:5d} " + " ".join(f"{k}={v:.3e}" for k, v in ld.items()), file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    mapper(input=None)  # step == max_num_iterations -> shut_down (snapshot)
    assert mapper.is_shut_down

    # ---- evaluation: held-out poses between keyframes, rendered at training resolution
    renderer = NerfstudioRenderer(mapping_model=mapper)
    intr = replica_intrinsics(height, width)
    intr_d = {"fx": intr[0], "fy": intr[1], "cx": intr[2], "cy": intr[3], "height": height, "width": width}
    all_poses = orbit_poses_opencv(keyframes * 2, device=dev)  # odd indices lie between keyframes
    ds = mapper.trainer.pipeline.datamanager.train_dataset
    norm = ds.normalization_matrix.to(dev)
    psnr_ref, psnr_flt, depth_l1 = [], [], []
    psnr_train, depth_l1_train = [], []
    for k in range(2 * eval_frames):
        held_out = k < eval_frames
        idx = (1 if held_out else 0) + 2 * int((k % eval_frames) * keyframes / eval_frames)
        pose_cv = all_poses[idx:idx + 1]
        color_gt, depth_gt, _ = render_room(pose_cv, height, width, intr)
        gt = (color_gt[0].permute(1, 2, 0).cpu().numpy() * 255).astype(np.uint8)
        # evaluation poses go through the same world normalisation the training poses got
        pose_gl = norm @ opencv_to_opengl(pose_cv)[0]
        pose_std = pose_gl.clone()
        pose_std[:3, 1:3] *= -1  # renderer expects the standard (OpenCV) convention
        color, depth = renderer.render_frame(intr_d, pose_std.cpu().numpy())
        if not held_out:  # keyframe poses: separates generalisation from fit
            psnr_train.append(calculate_psnr_float(color, gt))
            depth_l1_train.append(float(np.abs(depth - depth_gt[0, 0].cpu().numpy()).mean()))
            continue
        psnr_ref.append(calculate_psnr_reference(color, gt))
        psnr_flt.append(calculate_psnr_float(color, gt))
        depth_l1.append(float(np.abs(depth - depth_gt[0, 0].cpu().numpy()).mean()))
    res = {
        "keyframes": keyframes, "resolution": [width, height], "iterations": iterations,
        "train_seconds": train_s, "iterations_per_sec": iterations / train_s,
        "ray_samples_per_sec": iterations * 4096 * 48 / train_s,
        "psnr_reference_uint8wrap": float(np.mean(psnr_ref)), "psnr_float_mse": float(np.mean(psnr_flt)),
        "depth_l1": float(np.mean(depth_l1)), "psnr_float_mse_keyframe_views": float(np.mean(psnr_train)),
        "depth_l1_keyframe_views": float(np.mean(depth_l1_train)), "final_losses": mapper.trainer.pipeline.model.engine.loss_dict(),
        "snapshot_dir": out_dir, "deterministic": bool(deterministic), "dynamic_loss_scale": bool(mapper.trainer.pipeline.model.engine.cfg.dynamic_loss_scale),
        "seed": seed, "camera_optimizer_mode": camera_optimizer_mode or "SE3",
        "loss_scale_end": mapper.trainer.pipeline.model.engine.current_loss_scale(),
        "opt_steps": mapper.trainer.pipeline.model.engine.opt_steps,
        "pose_adjustment_rms": float(mapper.trainer.pipeline.model.engine.view("camera_opt.pose_adjustment").pow(2).mean().sqrt()),
    }
    if not quiet:
        print(json.dumps(res))
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--keyframes", type=int, default=48)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--iterations", type=int, default=1500)
    ap.add_argument("--eval-frames", type=int, default=6)
    ap.add_argument("--log-every", type=int, default=0)
    ap.add_argument("--deterministic", action="store_true")
    ap.add_argument("--dynamic-loss-scale", action="store_true", help="(the default since round 4: GradScaler, as the reference)")
    ap.add_argument("--static-loss-scale", action="store_true", help="tcnn's static loss scale 128 instead of GradScaler's dynamics")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--camera-optimizer-mode", default=None, help="SE3 (default) | SO3xR3 | off")
    a = ap.parse_args()
    run(a.keyframes, a.height, a.width, a.iterations, a.eval_frames, log_every=a.log_every, deterministic=a.deterministic,
        dynamic_loss_scale=False if a.static_loss_scale else None, seed=a.seed, camera_optimizer_mode=a.camera_optimizer_mode)
