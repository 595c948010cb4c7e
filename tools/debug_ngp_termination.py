#!/usr/bin/env python3
"""How many of the samples an inference bundle shades lie behind the point where the ray's transmittance has fallen
below the reference's render_min_transmittance (1e-4, evaluation/nerf_renderer.py:154)?  Trains the occupancy-grid
back-end on the synthetic room (as tools/ngp_bench.py), marches + shades one view and counts."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

entry.build()
from nerf_vo_amd import pyngp  # noqa: E402
from nerf_vo_amd.mapping.cameras import Cameras, CameraType  # noqa: E402
from nerf_vo_amd.mapping.dataset import opencv_to_opengl  # noqa: E402
from nerf_vo_amd.synthetic import make_sequence  # noqa: E402

dev = torch.device("cuda:0")
F, H, W = 48, 272, 480
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
    depths_cov=torch.ones_like(depth), resolution=np.array([W, H]),
    principal_point=seq["camera_intrinsics"][0, 2:].cpu().numpy(), focal_length=seq["camera_intrinsics"][0, :2].cpu().numpy())
for steps in (600, 2400):
    while tb.training_step < steps:
        tb.frame()
    eng = tb._engine
    gl = opencv_to_opengl(poses)[:, :3]
    fx, fy, cx, cy = [float(v) for v in seq["camera_intrinsics"][0]]
    cams = Cameras(fx=fx, fy=fy, cx=cx, cy=cy, height=H, width=W, camera_to_worlds=gl[5:6].contiguous(),
                   camera_type=CameraType.PERSPECTIVE).to(dev)
    b = cams.generate_rays(camera_indices=0, keep_shape=True)
    o, d, dn = b.origins.reshape(-1, 3)[:8192].contiguous(), b.directions.reshape(-1, 3)[:8192].contiguous(), \
        b.metadata["directions_norm"].reshape(-1)[:8192].contiguous()
    eng.render_rays(o, d, dn)
    ws = eng._wss[False]
    n = int(ws["offsets"][-1].item())
    ray = ws["ray_idx"][:n].long()
    sigma = torch.exp(ws["density_out"][:n, 0].float())
    tau = sigma * ws["dt"][:n]
    cum = torch.cumsum(tau.double(), 0)
    start = ws["offsets"][:-1].long()[ray]
    before = cum - tau.double() - torch.where(start > 0, cum[(start - 1).clamp_min(0)], torch.zeros_like(cum))
    dead = torch.exp(-before) < 1e-4
    cnt = ws["counts"][:8192].float()
    print(f"step {tb.training_step}: {n} samples of 8192 rays ({n / 8192:.1f} per ray, max {int(cnt.max())}); "
          f"{float(dead.float().mean()) * 100:.1f} % lie behind T < 1e-4; rays/batch {eng.rays_per_batch}; "
          f"occupied share of cascade 0: {float(torch.from_numpy(np.unpackbits(eng.bitfield.cpu().numpy()[: 128 ** 3 // 8])).float().mean()):.3f}",
          flush=True)
