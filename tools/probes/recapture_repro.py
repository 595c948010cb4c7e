"""Repeated drop-and-recapture of the step graphs of one engine (what tools/probes/dead_tiles_ab.py does between its
blocks): a crash here is a crash of the recapture flow, not of a feature under A/B."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nerf_vo_amd.engine import EngineConfig, NerfactoEngine  # noqa: E402
from nerf_vo_amd.mapping.dataset import DynamicDataset, opencv_to_opengl  # noqa: E402
from nerf_vo_amd.synthetic import make_sequence  # noqa: E402

dev = torch.device("cuda:0")
n, H, W, R = 24, 240, 320, 4096
seq = make_sequence(n, H, W, device=dev)
ds = DynamicDataset(num_frames=n, frame_height=H, frame_width=W, device=dev, use_normals=False)
ds.update({"keyframe_indices": torch.arange(n), "camera_intrinsics": seq["camera_intrinsics"],
           "camera_extrinsics": opencv_to_opengl(seq["camera_extrinsics"]), "frames_color": seq["frames_color"],
           "frames_depth": seq["frames_depth"]})
eng = NerfactoEngine(EngineConfig(num_images=n, num_rays=R), dev)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    eng._graphs.clear()
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 100):
        eng.train_step_graphed(ds)
    torch.cuda.synchronize()
    print("recapture", rep, "ok, step", eng.step, flush=True)
