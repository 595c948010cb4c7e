#!/usr/bin/env python3
"""Occupancy-grid back-end, training-batch compaction in rounds: samples per round, kept-count histogram and step time for
several first-round budgets at a late training stage.  python tools/debug_ngp_rounds.py [--warmup 5000]"""
import argparse
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
ap.add_argument("--warmup", type=int, default=5000)
ap.add_argument("--budgets", type=str, nargs="+", default=["", "32", "24,24", "16,16,32", "32,32", "24,40", "20,20,40"])
a = ap.parse_args()
dev = torch.device("cuda:0")
H, W, F = 272, 480, 48
seq = make_sequence(F, H, W, device=dev, scene_scale=0.2)
poses = seq["camera_extrinsics"].clone()
poses[:, :3, 3] += 0.5
tb = pyngp.Testbed(pyngp.TestbedMode.Nerf, 0)
tb.create_empty_nerf_dataset(n_images=F, nerf_scale=1.0, nerf_offset=np.zeros(3), aabb_scale=4)
tb.reload_network_from_file("")
tb.shall_train = True
tb.nerf.training.optimize_extrinsics = True
color = seq["frames_color"].permute(0, 2, 3, 1)
color = torch.cat([color, torch.ones_like(color[..., :1])], dim=3)
depth = seq["frames_depth"].permute(0, 2, 3, 1)
tb.nerf.training.update_training_images(
    frame_ids=list(range(F)), poses=opencv_to_opengl(poses)[:, :3], images=color.contiguous(), depths=depth.contiguous(),
    depths_cov=torch.ones_like(depth), resolution=np.array([W, H]), principal_point=seq["camera_intrinsics"][0, 2:].cpu().numpy(),
    focal_length=seq["camera_intrinsics"][0, :2].cpu().numpy())
for _ in range(a.warmup):
    tb.frame()
torch.cuda.synchronize()
eng = tb._engine
ws = eng._wss[True]
kept = ws["_per_ray"]["kept"][:eng.rays_per_batch].cpu().numpy()
print(f"step {eng.step}: rays/batch {eng.rays_per_batch}; kept per ray: mean {kept.mean():.1f}, percentiles 50/75/90/95/99 = "
      f"{np.percentile(kept, [50, 75, 90, 95, 99]).tolist()}, max {kept.max()}")
for k1 in a.budgets:
    eng.cfg.train_rounds = tuple(int(x) for x in k1.split(",") if x)
    for _ in range(48):
        tb.frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        tb.frame()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"rounds ({k1:>9s}, rest): {dt * 1e3:.3f} ms/step, rays {eng.rays_per_batch}, samples per round "
          f"{ws['totals_m'][:len(eng.cfg.train_rounds) + 1, 0].tolist()}, packed {ws['totals'].tolist()}", flush=True)
