#!/usr/bin/env python3
"""Loss trajectory of a long engine run on the synthetic room (stability check).
python tools/debug_longrun.py [--poses 0|1] [--acc 32|64] [--iters 8000] [--kf 48] [--h 120] [--w 160]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

entry.build()
from nerf_vo_amd.engine import EngineConfig, NerfactoEngine  # noqa: E402
from nerf_vo_amd.mapping.dataset import DynamicDataManagerConfig, opencv_to_opengl  # noqa: E402
from nerf_vo_amd.synthetic import make_sequence  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--poses", type=int, default=1)
    ap.add_argument("--acc", type=int, default=32)
    ap.add_argument("--iters", type=int, default=8000)
    ap.add_argument("--kf", type=int, default=48)
    ap.add_argument("--h", type=int, default=120)
    ap.add_argument("--w", type=int, default=160)
    ap.add_argument("--graph", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    dm = DynamicDataManagerConfig(train_num_rays_per_batch=4096, num_frames=a.kf, frame_height=a.h, frame_width=a.w,
                                  use_normals=False).setup(device=dev)
    seq = make_sequence(a.kf, a.h, a.w, device=dev)
    dm.train_dataset.update({"keyframe_indices": torch.arange(a.kf), "camera_intrinsics": seq["camera_intrinsics"],
                             "camera_extrinsics": opencv_to_opengl(seq["camera_extrinsics"]),
                             "frames_color": seq["frames_color"], "frames_depth": seq["frames_depth"]})
    ds = dm.train_dataset
    cfg = EngineConfig(num_images=a.kf, optimize_poses=bool(a.poses), max_num_iterations=a.iters)
    eng = NerfactoEngine(cfg, dev)
    scale = torch.tensor([a.kf, a.h, a.w], device=dev)
    for it in range(a.iters):
        if a.graph:
            eng.train_step_graphed(ds)
        else:
            idx = torch.floor(torch.rand(4096, 3, device=dev) * scale).long()
            eng.train_step(idx, ds.camera_intrinsics, ds.camera_extrinsics[:, :3, :4].contiguous(), ds.frames_color, ds.frames_depth)
        if it % 500 == 0 or it == a.iters - 1:
            ld = eng.loss_dict()
            extra = ""
            if a.poses:
                pa = eng.view("camera_opt.pose_adjustment").view(a.kf, 6)
                extra = f" |pose_adj| max {pa.abs().max().item():.4f}"
            print(f"it {it:5d} " + " ".join(f"{k}={v:.3e}" for k, v in ld.items()) + f" skip={int(eng.skip_flag.sum().item())}" + extra, flush=True)


if __name__ == "__main__":
    main()
