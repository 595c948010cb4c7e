"""Are steps 5..24 of a fresh field slow because of the FIELD (untrained densities) or because of the GPU (clocks / caches
after the capture pause)?  Engine B is timed exactly as the driver does (5 warm-up steps, 20 timed) -- once on a cold
GPU, once right behind 400 steps of another engine A."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nerf_vo_amd.engine import EngineConfig, NerfactoEngine  # noqa: E402
from nerf_vo_amd.mapping.dataset import DynamicDataset, opencv_to_opengl  # noqa: E402
from nerf_vo_amd.synthetic import make_sequence  # noqa: E402

dev = torch.device("cuda:0")
n, H, W, R = 192, 480, 640, 4096
seq = make_sequence(n, H, W, device=dev)
ds = DynamicDataset(num_frames=n, frame_height=H, frame_width=W, device=dev, use_normals=False)
ds.update({"keyframe_indices": torch.arange(n), "camera_intrinsics": seq["camera_intrinsics"],
           "camera_extrinsics": opencv_to_opengl(seq["camera_extrinsics"]), "frames_color": seq["frames_color"],
           "frames_depth": seq["frames_depth"]})


def timed(eng, warm, steps):
    for _ in range(warm):
        eng.train_step_graphed(ds)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.train_step_graphed(ds)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


torch.manual_seed(0)
b = NerfactoEngine(EngineConfig(num_images=n, num_rays=R), dev)
print(f"fresh engine, cold GPU:         steps 5..24: {timed(b, 5, 20):.4f} ms/step, steps 25..44: {timed(b, 0, 20):.4f}")
a = NerfactoEngine(EngineConfig(num_images=n, num_rays=R), dev)
print(f"another engine, 400 steps:      {timed(a, 20, 380):.4f} ms/step")
torch.manual_seed(0)
c = NerfactoEngine(EngineConfig(num_images=n, num_rays=R), dev)
c.train_step_graphed(ds)  # (captures; one step)
timed(a, 0, 200)          # keep the GPU busy right up to the measurement
print(f"fresh engine behind a busy GPU: steps 1..5 warm-up, steps 6..25: {timed(c, 4, 20):.4f} ms/step, next 20: {timed(c, 0, 20):.4f}")
