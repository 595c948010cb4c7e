import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry; entry.build()
from nerf_vo_amd.engine import EngineConfig, NerfactoEngine
from nerf_vo_amd.mapping.dataset import DynamicDataset, opencv_to_opengl
from nerf_vo_amd.synthetic import make_sequence
dev = torch.device("cuda:0")
n, H, W = 24, 120, 160
seq = make_sequence(n, H, W, device=dev)
def run(tag, graph, **kw):
    ds = DynamicDataset(num_frames=n, frame_height=H, frame_width=W, device=dev, use_normals=False)
    ds.update({"keyframe_indices": torch.arange(n), "camera_intrinsics": seq["camera_intrinsics"],
               "camera_extrinsics": opencv_to_opengl(seq["camera_extrinsics"]), "frames_color": seq["frames_color"],
               "frames_depth": seq["frames_depth"]})
    torch.manual_seed(0)
    eng = NerfactoEngine(EngineConfig(num_images=n, **kw), dev)
    c2w = ds.camera_extrinsics[:, :3, :4].contiguous()
    out = []
    for it in range(1500):
        if graph:
            eng.train_step_graphed(ds)
        else:
            idx = torch.floor(torch.rand(4096, 3, device=dev) * torch.tensor([n, H, W], device=dev)).long()
            eng.train_step(idx, ds.camera_intrinsics, c2w, ds.frames_color, ds.frames_depth)
        if it % 300 == 299:
            p = eng.view("camera_opt.pose_adjustment").view(n, 6)
            out.append((it + 1, round(eng.loss_dict()["rgb_loss"], 5), float(p[:, :3].norm(dim=1).mean()), float(p[:, 3:].norm(dim=1).mean()), int(eng.skip_flag.item())))
    print(tag, out)
run("fixed eager", False)
run("poses eager", False, optimize_poses=True)
run("poses eager lr0", False, optimize_poses=True, lr_camera=1e-12, lr_camera_final=1e-12)
run("poses graph", True, optimize_poses=True)
run("fixed graph", True)
