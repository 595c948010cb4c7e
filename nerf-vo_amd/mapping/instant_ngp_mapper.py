"""Drop-in counterpart of the reference's second mapping method ``InstantNGP``
(/root/reference/nerf_vo/mapping/instant_ngp.py:19-117) and of ``InstantNGPRenderer``
(/root/reference/evaluation/nerf_renderer.py:221-319), over the native occupancy-grid engine
(nerf_vo_amd.ngp_engine) instead of pyngp.Testbed.

Same constructor, attributes (``is_initialized``, ``is_shut_down``, ``step``) and methods
(``__call__``, ``update``, ``train``, ``shut_down``, ``save_snapshot``); same ingest transformations
(NCHW -> NHWC, sRGB -> linear colours at :64-67, unit depth covariance, intrinsics of the first frame,
poses taken as camera-to-world 3x4 with nerf_scale 1 / nerf_offset 0, aabb_scale 4).  Differences:
keyframe data stays on the device (the reference round-trips through host numpy lists, :87-100);
snapshots are torch files, not .msgpack; extrinsics optimisation inside the NGP trainer
(optimize_extrinsics, :47) is not built yet."""
from __future__ import annotations

import argparse
import math
from pathlib import Path

import numpy as np
import torch

from ..ngp_engine import NgpConfig, NgpEngine
from .cameras import Cameras, CameraType
from .nerfstudio_mapper import step_check
from .renderer import NeRFRenderer


class InstantNGP:
    def __init__(self, args: argparse.Namespace, device: torch.device = torch.device("cuda:0")) -> None:
        self.args = args
        self.device = torch.device(device)
        self.is_initialized = False
        self.is_shut_down = False
        self.step = 0
        n, h, w = args.num_keyframes, args.frame_height, args.frame_width
        self.ngp = NgpEngine(NgpConfig(num_images=n, aabb_scale=4, depth_loss_mult=1.0), self.device)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.images = torch.zeros(n, h, w, 3, **f32)          # linear colours
        self.depths = torch.zeros(n, h, w, 1, **f32)
        self.depths_cov = torch.ones(n, h, w, 1, **f32)
        self.poses = torch.eye(4, **f32)[:3].repeat(n, 1, 1)  # camera-to-world, OpenGL axes
        self.intrinsics = torch.zeros(n, 4, **f32)
        self.n_images_for_training = 0
        self.generator = torch.Generator(device=self.device)
        self.generator.manual_seed(42)

    def __call__(self, input: dict | None) -> None:
        if self.step == self.args.mapping_iterations:
            self.shut_down()
        else:
            if input is not None:
                self.update(input=input)
            if self.is_initialized:
                self.train()

    def update(self, input: dict) -> None:
        idx = input["keyframe_indices"].to(self.device).long()
        color = input["frames_color"].to(self.device).permute(0, 2, 3, 1)
        # sRGB -> linear, exactly as the reference does before handing images to the testbed
        color = torch.where(color > 0.04045, torch.pow((color + 0.055) / 1.055, 2.4), color / 12.92)
        pose = input["camera_extrinsics"].to(self.device)[:, :3].clone()
        pose[:, :3, 1:3] *= -1  # OpenCV camera axes -> OpenGL (what the native ray generator expects)
        self.images[idx] = color
        self.depths[idx] = input["frames_depth"].to(self.device).permute(0, 2, 3, 1)
        if "frames_depth_covariance" in input:
            self.depths_cov[idx] = input["frames_depth_covariance"].to(self.device).permute(0, 2, 3, 1)
        self.poses[idx] = pose
        # the reference passes ONE focal length / principal point (first frame of the packet)
        self.intrinsics[idx] = input["camera_intrinsics"].to(self.device)[0]
        self.n_images_for_training = max(self.n_images_for_training, int(idx.max()) + 1)
        self.is_initialized = True

    def train(self) -> None:
        n, h, w = self.n_images_for_training, self.args.frame_height, self.args.frame_width
        scale = torch.tensor([n, h, w], device=self.device)
        u = torch.rand((self.ngp.cfg.num_rays, 3), device=self.device, generator=self.generator)
        ray_indices = torch.floor(u * scale).long()
        self.ngp.train_step(ray_indices, self.intrinsics, self.poses, self.images, self.depths)
        if step_check(self.step, self.args.mapping_snapshot_iterations):
            self.save_snapshot()
        self.step += 1

    def shut_down(self) -> None:
        self.save_snapshot()
        self.is_shut_down = True

    def save_snapshot(self) -> None:
        d = Path(self.args.dir_prediction) / "snapshots"
        d.mkdir(parents=True, exist_ok=True)
        e = self.ngp
        torch.save({"step": self.step, "params": e.params, "exp_avg": e.exp_avg, "exp_avg_sq": e.exp_avg_sq,
                    "opt_step": e.opt_step, "density_grid": e.density_grid, "bitfield": e.bitfield,
                    "poses": self.poses[: self.n_images_for_training], "config": vars(e.cfg)},
                   d / f"snapshot{self.step:06d}.pt")

    def get_camera_extrinsics(self, frame_idx: int) -> np.ndarray:
        """3x4 camera-to-world of a training frame (OpenGL axes), the testbed call the renderer uses."""
        return self.poses[frame_idx].detach().cpu().numpy()


class InstantNGPRenderer(NeRFRenderer):
    """render_frame(intrinsics, extrinsics) -> (uint8 sRGB colour, depth) like the reference: linear
    render -> sRGB transfer -> clip -> *255 + 0.5."""

    def load_nerf_from_snapshot(self, dir_prediction: str) -> None:
        raise NotImplementedError("offline snapshot reload is a 'next' row (SURVEY.md section 8f, f4)")

    def load_nerf_from_mapping_model(self, mapping_model) -> None:
        self.mapper = mapping_model
        self.ngp = mapping_model.ngp

    def get_camera_extrinsics(self, frame_index: int) -> np.ndarray:
        m = np.eye(4)
        m[:3] = self.mapper.get_camera_extrinsics(frame_index)
        m[0:3, 1:3] *= -1  # OpenGL -> standard convention
        return m

    def render_frame(self, camera_intrinsics: dict, camera_extrinsics: np.ndarray, rays_per_chunk: int = 2048):
        ext = np.array(camera_extrinsics, dtype=np.float64, copy=True)
        ext[0:3, 1:3] *= -1  # standard -> OpenGL
        dev = self.ngp.device
        cams = Cameras(fx=camera_intrinsics["fx"], fy=camera_intrinsics["fy"], cx=camera_intrinsics["cx"],
                       cy=camera_intrinsics["cy"], height=camera_intrinsics["height"], width=camera_intrinsics["width"],
                       camera_to_worlds=torch.tensor(ext, dtype=torch.float32).unsqueeze(0)[:, :3],
                       camera_type=CameraType.PERSPECTIVE).to(dev)
        bundle = cams.generate_rays(camera_indices=0, keep_shape=True)
        H, W = camera_intrinsics["height"], camera_intrinsics["width"]
        o = bundle.origins.reshape(-1, 3)
        d = bundle.directions.reshape(-1, 3)
        dn = bundle.metadata["directions_norm"].reshape(-1)
        rgb, depth = [], []
        for lo in range(0, o.shape[0], rays_per_chunk):
            hi = min(o.shape[0], lo + rays_per_chunk)
            oo, dd, nn = o[lo:hi], d[lo:hi], dn[lo:hi]
            if hi - lo < rays_per_chunk:
                pad = rays_per_chunk - (hi - lo)
                oo = torch.cat([oo, oo[-1:].expand(pad, 3)])
                dd = torch.cat([dd, dd[-1:].expand(pad, 3)])
                nn = torch.cat([nn, nn[-1:].expand(pad)])
            out = self.ngp.render_rays(oo.contiguous(), dd.contiguous(), nn.contiguous())
            rgb.append(out["rgb"][: hi - lo].clone())
            depth.append((out["depth"][: hi - lo, 0] / nn[: hi - lo]).clone())  # ray distance -> z-depth
        lin = torch.cat(rgb).view(H, W, 3).cpu().numpy()
        srgb = np.where(lin > 0.0031308, 1.055 * (np.maximum(lin, 1e-12) ** (1.0 / 2.4)) - 0.055, 12.92 * lin)
        color = (np.clip(srgb, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
        return color, torch.cat(depth).view(H, W).cpu().numpy()
