#!/usr/bin/env python3
"""Where does the train-loss / eval-render PSNR gap come from?  Trains on the synthetic room, then renders
keyframe views with (a) the mean appearance embedding (nerfacto eval semantics) and (b) the keyframe's own
embedding (what the training loss sees)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

entry.build()
from nerf_vo_amd.engine import EngineConfig, NerfactoEngine  # noqa: E402
from nerf_vo_amd.mapping.dataset import DynamicDataManagerConfig, opencv_to_opengl  # noqa: E402
from nerf_vo_amd.synthetic import make_sequence  # noqa: E402


def psnr(a, b):
    return float(-10 * torch.log10(((a - b) ** 2).mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=4000)
    ap.add_argument("--kf", type=int, default=48)
    ap.add_argument("--h", type=int, default=120)
    ap.add_argument("--w", type=int, default=160)
    ap.add_argument("--poses", type=int, default=1)
    ap.add_argument("--loss-scale", type=float, default=None)
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
    extra = {} if a.loss_scale is None else {"loss_scale": a.loss_scale}
    eng = NerfactoEngine(EngineConfig(num_images=a.kf, optimize_poses=bool(a.poses), max_num_iterations=a.iters, **extra), dev)
    for it in range(a.iters):
        eng.train_step_graphed(ds)
    print("final train losses", eng.loss_dict(), flush=True)
    emb = eng.view("field.embedding").view(a.kf, -1)
    print("embedding rms", float(emb.pow(2).mean().sqrt()), "mean-embedding rms", float(emb.mean(0).pow(2).mean().sqrt()))
    mean_emb = emb.mean(dim=0, keepdim=True).to(torch.float16).contiguous()
    for k in (3, 17, 40):
        bundle = ds.cameras.generate_rays(camera_indices=k, keep_shape=True)
        o = bundle.origins.reshape(-1, 3).contiguous()
        d = bundle.directions.reshape(-1, 3).contiguous()
        dn = bundle.metadata["directions_norm"].reshape(-1).contiguous()
        gt = ds.frames_color[k].reshape(-1, 3).float()
        if gt.max() > 2:
            gt = gt / 255.0
        gtd = ds.frames_depth[k].reshape(-1).float()
        res = {}
        for name, e in (("mean", mean_emb), ("own", emb[k:k + 1].to(torch.float16).contiguous()),
                        ("zero", torch.zeros_like(mean_emb))):
            r = eng.render_rays(o, d, dn, e)
            z = (r["depth"].reshape(-1) / dn)
            res[name] = (psnr(r["rgb"].reshape(-1, 3), gt), float((z - gtd).abs().mean()), float(r["accumulation"].mean()))
        # training-mode forward on the same full image (4096-ray tiles, jitter 0.5 = bin centres)
        H, W = a.h, a.w
        ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
        full = torch.stack([torch.full_like(ys, k), ys, xs], dim=-1).reshape(-1, 3).long()
        c2w = ds.camera_extrinsics[:, :3, :4].contiguous()
        ws = eng._workspace(4096, True)
        tl, n_t, dmax, omax, tl_c = 0.0, 0, 0.0, 0.0, 0.0
        for lo in range(0, full.shape[0] - 4095, 4096):
            idx = full[lo:lo + 4096].contiguous()
            eng.load_rays(ws, idx, ds.camera_intrinsics, c2w, ds.frames_color, ds.frames_depth)
            dmax = max(dmax, float((ws["directions"] - d[lo:lo + 4096]).abs().max()))
            omax = max(omax, float((ws["origins"] - o[lo:lo + 4096]).abs().max()))
            jit = tuple(torch.rand(4096, device=dev) for _ in range(3))
            eng.forward_backward(ws, jit, has_depth=True, update_proposals=False)
            tl += eng.loss_dict()["rgb_loss"]
            jit = tuple(torch.full((4096,), 0.5, device=dev) for _ in range(3))
            eng.forward_backward(ws, jit, has_depth=True, update_proposals=False)
            tl_c += eng.loss_dict()["rgb_loss"]
            n_t += 1
        print(f"   training-mode rgb loss over the image: {tl / n_t:.3e} (psnr {-10 * np.log10(tl / n_t):.2f}), "
              f"with jitter 0.5: {tl_c / n_t:.3e} (psnr {-10 * np.log10(tl_c / n_t):.2f});"
              f" raygen vs Cameras.generate_rays: max|d dir| {dmax:.2e} max|d origin| {omax:.2e}")
        # sensitivity to a small camera-local rotation (what a pose correction of that size does)
        from nerf_vo_amd.synthetic import render_room, replica_intrinsics
        for ang in (5e-4, 2e-3):
            rot = torch.eye(3, device=dev)
            rot[0, 0] = rot[2, 2] = float(np.cos(ang))
            rot[0, 2] = float(np.sin(ang))
            rot[2, 0] = -float(np.sin(ang))
            c2w_k = ds.camera_extrinsics[k, :3, :3]
            d_p = (d @ c2w_k) @ rot.T @ c2w_k.T  # rotate in the camera frame
            r0 = eng.render_rays(o, d, dn, emb[k:k + 1].to(torch.float16).contiguous())["rgb"].reshape(-1, 3).clone()
            r1 = eng.render_rays(o, d_p.contiguous(), dn, emb[k:k + 1].to(torch.float16).contiguous())["rgb"].reshape(-1, 3)
            pose_cv = seq["camera_extrinsics"][k:k + 1].clone()
            pose_p = pose_cv.clone()
            pose_p[0, :3, :3] = pose_cv[0, :3, :3] @ rot
            intr = replica_intrinsics(a.h, a.w)
            g0 = render_room(pose_cv, a.h, a.w, intr)[0][0].permute(1, 2, 0).reshape(-1, 3)
            g1 = render_room(pose_p, a.h, a.w, intr)[0][0].permute(1, 2, 0).reshape(-1, 3)
            print(f"   rotation {ang:.0e} rad: model(rot) vs model {psnr(r1, r0):.2f} dB, GT(rot) vs GT {psnr(g1, g0):.2f} dB,"
                  f" model(rot) vs GT {psnr(r1, gt):.2f} dB")
        print(f"keyframe {k}: " + "  ".join(f"{n}: psnr {v[0]:.2f} depthL1 {v[1]:.3f} acc {v[2]:.3f}" for n, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
