#!/usr/bin/env python3
"""PSNR of the instant-ngp mapper mirror's end-to-end flow (tests/test_mapping_gpu.py) with / without the untrained-cell
marking, a few runs each (training is non-deterministic: float atomics)."""
import argparse
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

entry.build()
from nerf_vo_amd import ngp_engine, pyngp  # noqa: E402
from nerf_vo_amd.mapping.dataset import opencv_to_opengl  # noqa: E402
from nerf_vo_amd.mapping.instant_ngp_mapper import InstantNGP, InstantNGPRenderer  # noqa: E402
from nerf_vo_amd.mapping.renderer import calculate_psnr_float  # noqa: E402
from nerf_vo_amd.synthetic import make_sequence, replica_intrinsics  # noqa: E402

dev = torch.device("cuda:0")
n, H, W, iters = 12, 68, 120, 400
seq = make_sequence(n, H, W, device=dev, scene_scale=0.2)
poses = seq["camera_extrinsics"].clone()
poses[:, :3, 3] += 0.5
_tinit = pyngp._Training.__init__
for mark, margin, warm, rbg in ((True, 1.0, 256, False), (True, 1.0, 256, True)):
    def tpatched(self, testbed, _r=rbg):
        _tinit(self, testbed)
        self.random_bg_color = _r
    pyngp._Training.__init__ = tpatched
    res = []
    for run in range(3):
        with tempfile.TemporaryDirectory() as tmp:
            args = argparse.Namespace(num_keyframes=n, frame_height=H, frame_width=W, mapping_iterations=iters,
                                      mapping_snapshot_iterations=iters, dir_prediction=tmp)
            mapper = InstantNGP(args, device=dev)
            mapper(input={"keyframe_indices": torch.arange(n), "camera_intrinsics": seq["camera_intrinsics"],
                          "camera_extrinsics": opencv_to_opengl(poses), "frames_color": seq["frames_color"],
                          "frames_depth": seq["frames_depth"], "last_frame": True})
            while mapper.step < iters:
                mapper(input=None)
            mapper(input=None)
            renderer = InstantNGPRenderer(mapping_model=mapper)
            fx, fy, cx, cy = replica_intrinsics(H, W)
            intr = {"fx": fx, "fy": fy, "cx": cx, "cy": cy, "height": H, "width": W}
            ps = []
            for f in (1, 3, 6, 10):
                color, depth = renderer.render_frame(intr, renderer.get_camera_extrinsics(f))
                gt = (seq["frames_color"][f].permute(1, 2, 0).cpu().numpy() * 255).astype(np.uint8)
                ps.append(calculate_psnr_float(color, gt))
            eng = mapper.ngp._engine
            bits = torch.from_numpy(np.unpackbits(eng.bitfield.cpu().numpy(), bitorder="little").astype(bool)).to(dev)
            g3 = eng.density_grid.view(3, -1)
            b3 = bits.view(3, -1)
            res.append((round(ps[1], 2), round(float(np.mean(ps)), 2), eng.rays_per_batch,
                        [round(float((g3[l] < 0).float().mean()), 3) for l in range(3)],
                        [round(float(b3[l].float().mean()), 4) for l in range(3)],
                        round(float(g3[0].clamp_min(0).mean()), 5), eng.loss_dict()))
    print(f"mark_untrained={mark} margin={margin} density_warmup_steps={warm} random_bg={rbg}: (psnr frame 3, mean of 4 frames, rays/batch, unseen share per cascade, occupied share per cascade, mean of cascade 0, losses) {res}", flush=True)
