"""Fraction of proposal samples whose dL/d(density pre-activation) is exactly zero in the 16-bit buffer the proposal
backward reads, under the static loss scale (tcnn: 128) and under GradScaler's dynamic scale (init 65536) -- what decides
how many samples the proposal grids' slice-owner scan can skip, and which gradients Adam ever sees."""
import argparse
import json
import sys

import torch

sys.path.insert(0, ".")
from nerf_vo_amd.engine import EngineConfig, NerfactoEngine  # noqa: E402
from nerf_vo_amd.mapping.dataset import DynamicDataset, opencv_to_opengl  # noqa: E402
from nerf_vo_amd.synthetic import make_sequence  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keyframes", type=int, default=48)
    ap.add_argument("--steps", type=int, nargs="+", default=[0, 8, 100, 1000])
    ap.add_argument("--mlp-dtype", default="f16")
    args = ap.parse_args()
    device = torch.device("cuda:0")
    n, H, W = args.keyframes, 240, 320
    seq = make_sequence(n, H, W, device=device)
    for dynamic in (False, True):
        torch.manual_seed(1)
        ds = DynamicDataset(num_frames=n, frame_height=H, frame_width=W, device=device, use_normals=False)
        ds.update({"keyframe_indices": torch.arange(n), "camera_intrinsics": seq["camera_intrinsics"],
                   "camera_extrinsics": opencv_to_opengl(seq["camera_extrinsics"]), "frames_color": seq["frames_color"],
                   "frames_depth": seq["frames_depth"]})
        eng = NerfactoEngine(EngineConfig(num_images=n, num_rays=4096, dynamic_loss_scale=dynamic, mlp_dtype=args.mlp_dtype), device)
        c2w = ds.camera_extrinsics[:, :3, :4].contiguous()
        gen = torch.Generator(device=device).manual_seed(5)
        out = {"dynamic_loss_scale": dynamic, "fractions": {}}
        last = max(args.steps)
        for step in range(last + 1):
            if step in args.steps:
                # an eager step whose proposal networks train: the buffers of this step stay readable afterwards
                idx = torch.floor(torch.rand(4096, 3, device=device, generator=gen) * torch.tensor([n, H, W], device=device)).long()
                ws = eng._workspace(4096, True)
                eng.load_rays(ws, idx, ds.camera_intrinsics, c2w, ds.frames_color, ds.frames_depth)
                jit = tuple(torch.rand(4096, device=device, generator=gen) for _ in range(3))
                eng.forward_backward(ws, jit, has_depth=True, update_proposals=True)
                torch.cuda.synchronize()
                fr = [float((ws[f"dout{k}"].view(torch.int16).reshape(-1)[: 4096 * eng.levels[k]] << 1 == 0).float().mean()) for k in range(2)]
                out["fractions"][step] = {"level0": round(fr[0], 4), "level1": round(fr[1], 4), "loss_scale": eng.current_loss_scale()}
            eng.train_step_graphed(ds)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
