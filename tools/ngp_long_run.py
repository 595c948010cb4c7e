#!/usr/bin/env python3
"""Long run of the occupancy-grid back-end on the bench scene (48 keyframes 480x272): losses, applied vs attempted
optimiser steps, ray batch, occupancy and the PSNR of views between the training cameras every `--every` steps.
python tools/ngp_long_run.py [--steps 8000] [--every 1000]"""
import argparse
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

entry.build()
from nerf_vo_amd import pyngp  # noqa: E402
from nerf_vo_amd.mapping.dataset import opencv_to_opengl  # noqa: E402
from nerf_vo_amd.synthetic import make_sequence  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=8000)
ap.add_argument("--every", type=int, default=1000)
ap.add_argument("--keyframes", type=int, default=48)
ap.add_argument("--extrinsics", type=int, default=1)
ap.add_argument("--random-bg", type=int, default=1)
ap.add_argument("--depth-lambda", type=float, default=1.0)
ap.add_argument("--ema", type=float, default=0.95)
a = ap.parse_args()
dev = torch.device("cuda:0")
H, W, F = 272, 480, a.keyframes
dense = make_sequence(2 * F, H, W, device=dev, scene_scale=0.2)
dposes = dense["camera_extrinsics"].clone()
dposes[:, :3, 3] += 0.5
poses = dposes[::2].contiguous()
tb = pyngp.Testbed(pyngp.TestbedMode.Nerf, 0)
tb.create_empty_nerf_dataset(n_images=F, nerf_scale=1.0, nerf_offset=np.zeros(3), aabb_scale=4)
tb.reload_network_from_file("")
tb.shall_train = True
tb.nerf.training.optimize_extrinsics = bool(a.extrinsics)
tb.nerf.training.random_bg_color = bool(a.random_bg)
tb.nerf.training.depth_supervision_lambda = a.depth_lambda
color = dense["frames_color"][::2].permute(0, 2, 3, 1)
color = torch.cat([color, torch.ones_like(color[..., :1])], dim=3)
depth = dense["frames_depth"][::2].permute(0, 2, 3, 1)
tb.nerf.training.update_training_images(
    frame_ids=list(range(F)), poses=opencv_to_opengl(poses)[:, :3], images=color.contiguous(), depths=depth.contiguous(),
    depths_cov=torch.ones_like(depth), resolution=np.array([W, H]),
    principal_point=dense["camera_intrinsics"][0, 2:].cpu().numpy(), focal_length=dense["camera_intrinsics"][0, :2].cpu().numpy())


def view_psnr(i):
    if i % 2 == 0:  # a training view: at the model's own (optimised) pose, as the reference's evaluation renders keyframes
        tb.set_nerf_camera_matrix(tb.nerf.training.get_camera_extrinsics(i // 2))
    else:           # a view between two training cameras: at its dataset pose
        mm = dposes[i].detach().cpu().numpy().astype(np.float64).copy()
        mm[0:3, 1:3] *= -1
        tb.set_nerf_camera_matrix(mm[[2, 0, 1]])
    tb.fov_axis, tb.fov = 0, 2.0 * math.degrees(math.atan(0.5 * W / float(dense["camera_intrinsics"][0, 0])))
    tb.render_mode = pyngp.Shade
    img = np.clip(tb.render(width=W, height=H, spp=1, linear=True)[..., :3], 0.0, 1.0)
    mse = float(np.mean((img - dense["frames_color"][i].permute(1, 2, 0).cpu().numpy()) ** 2))
    return 10.0 * math.log10(1.0 / max(mse, 1e-12))


t0 = time.perf_counter()
while tb.training_step < a.steps:
    tb.frame()
    if tb.training_step % a.every == 0:
        eng = tb._engine
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        held = float(np.mean([view_psnr(i) for i in range(1, 2 * F, 2 * F // 4)]))
        seen = float(np.mean([view_psnr(i) for i in range(0, 2 * F, 2 * F // 4)]))
        occ = float(torch.from_numpy(np.unpackbits(eng.bitfield.cpu().numpy()[: 128 ** 3 // 8])).float().mean())
        ld = eng.loss_dict()
        print(f"step {tb.training_step}: {el:.1f} s, rgb {ld['rgb_loss']:.2e} depth {ld['depth_loss']:.2e}, applied "
              f"{eng.applied_steps}/{eng.opt_step}, rays/batch {eng.rays_per_batch}, cascade-0 occupancy {occ:.3f}, captures "
              f"{eng.graph_captures}, PSNR training views {seen:.2f} dB / between {held:.2f} dB, pose offset rms "
              f"{float(eng.pose_adjustment.pow(2).mean().sqrt()):.2e}", flush=True)
        t0 = time.perf_counter() - el
