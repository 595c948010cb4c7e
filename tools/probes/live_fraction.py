"""How much of the main level carries a gradient on the driver-style workload (192 keyframes 640x480, 4096 rays, random
init): fraction of 16-sample tiles with a non-zero byte, of samples with a non-zero dL/doutput row of the base network,
of samples with a non-zero dL/d(rgb), at a few steps of the schedule."""
import os
import sys

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
eng = NerfactoEngine(EngineConfig(num_images=n, num_rays=R), dev)
done = 0
for upto in (20, 60, 120, 220, 500, 1000, 2000):
    while done < upto:
        eng.train_step_graphed(ds)
        done += 1
    torch.cuda.synchronize()
    ws = eng._workspace(R, True)
    live = ws["tile_live"]
    rows = ws["dout2"].float().abs().sum(1) != 0
    rgb = ws["drgb"].float().abs().sum(1) != 0
    print(f"step {done:5d}: loss scale {eng.current_loss_scale():g}  tiles live {float((live != 0).float().mean()):.3f}  "
          f"rows live {float(rows.float().mean()):.3f}  rgb rows live {float(rgb.float().mean()):.3f}", flush=True)
