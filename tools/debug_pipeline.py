"""Find the first step at which the pipelined and the plain single-GPU graphs differ (deterministic mode)."""
import sys
import torch
sys.path.insert(0, ".")
from nerf_vo_amd.engine import EngineConfig, NerfactoEngine
from nerf_vo_amd.mapping.dataset import DynamicDataset, opencv_to_opengl
from nerf_vo_amd.synthetic import make_sequence

device = torch.device("cuda:0")
n, H, W, R = 8, 60, 80, 512
seq = make_sequence(n, H, W, device=device)
poses = "--no-poses" not in sys.argv
ing = "--no-ingest" not in sys.argv
eager = "--no-eager" not in sys.argv


def ingest(ds, lo, hi):
    ds.update({"keyframe_indices": torch.arange(lo, hi), "camera_intrinsics": seq["camera_intrinsics"][lo:hi],
               "camera_extrinsics": opencv_to_opengl(seq["camera_extrinsics"][lo:hi]),
               "frames_color": seq["frames_color"][lo:hi], "frames_depth": seq["frames_depth"][lo:hi]})


def run(pipeline):
    torch.manual_seed(3)
    ds = DynamicDataset(num_frames=n, frame_height=H, frame_width=W, device=device, use_normals=False)
    ingest(ds, 0, 5 if ing else 8)
    eng = NerfactoEngine(EngineConfig(num_images=n, num_rays=R, optimize_poses=poses, deterministic=True,
                                      pipeline_sampling_prefix=pipeline), device)
    gen = torch.Generator(device=device).manual_seed(9)
    out = []
    for it in range(26):
        if ing and it == 14:
            ingest(ds, 5, 8)
        if eager and it == 20:
            extent = torch.tensor([ds.num_active_frames, H, W], device=device)
            idx = torch.floor(torch.rand(R, 3, device=device, generator=gen) * extent).long()
            eng.train_step(idx, ds.camera_intrinsics, ds.camera_extrinsics[:, :3, :4].contiguous(), ds.frames_color, ds.frames_depth)
        eng.train_step_graphed(ds)
        out.append((eng.loss_totals().clone(), eng.params.clone()))
    torch.cuda.synchronize()
    return out, eng


a, ea = run(True)
b, eb = run(False)
for it, ((la, pa), (lb, pb)) in enumerate(zip(a, b)):
    same_l = torch.equal(la, lb)
    same_p = torch.equal(pa, pb)
    if not same_p:
        print("first difference at step", it, "losses equal:", same_l, "params equal:", same_p)
        print(la.tolist()); print(lb.tolist())
        for g, (lo, hi) in ea.group_ranges.items():
            print(g, int((pa[lo:hi] != pb[lo:hi]).sum()), "of", hi - lo)
        break
else:
    print("identical over", len(a), "steps")
