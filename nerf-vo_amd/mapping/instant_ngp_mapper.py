"""Drop-in counterpart of the reference's second mapping method ``InstantNGP``
(/root/reference/nerf_vo/mapping/instant_ngp.py:19-117) and of ``InstantNGPRenderer`` /
``NeRFSLAMNGPRenderer`` (/root/reference/evaluation/nerf_renderer.py:221-344).  Both drive the testbed facade
``nerf_vo_amd.pyngp`` through exactly the calls the reference makes on NVlabs' ``pyngp`` -- the facade, not these
classes, is the drop-in boundary: the reference's own two files run against it unchanged.

Same constructor, attributes (``is_initialized``, ``is_shut_down``, ``step``, ``ngp``) and methods
(``__call__``, ``update``, ``train``, ``shut_down``, ``save_snapshot``); same ingest transformations
(NCHW -> NHWC, sRGB -> linear colours + unit alpha at :64-74, unit depth covariance, intrinsics of the first
frame of the packet, poses taken as camera-to-world 3x4 in the NeRF / OpenGL convention the enhancement stage
emits, nerf_scale 1 / nerf_offset 0, aabb_scale 4, extrinsics optimisation on, L2 depth loss).  One difference:
keyframe data is handed to the facade as device tensors (the reference round-trips every packet through host
numpy lists, :87-100; the facade accepts both)."""
from __future__ import annotations

import argparse
import math
import os

import numpy as np
import torch

from .. import pyngp
from .nerfstudio_mapper import step_check
from .renderer import NeRFRenderer

file_instant_ngp_config = "nerf_vo/thirdparty/nerf_slam/thirdparty/instant-ngp/configs/nerf/base.json"


class InstantNGP:
    def __init__(self, args: argparse.Namespace, device: torch.device = torch.device("cuda:0")) -> None:
        self.args = args
        self.device = torch.device(device)
        self.is_initialized = False
        self.is_shut_down = False
        self.step = 0

        self.ngp = pyngp.Testbed(pyngp.TestbedMode.Nerf, self.device.index or 0)
        bounding_box = pyngp.BoundingBox(np.array([-np.inf, -np.inf, -np.inf]), np.array([np.inf, np.inf, np.inf]))
        self.ngp.create_empty_nerf_dataset(n_images=args.num_keyframes, nerf_scale=1.0,
                                           nerf_offset=np.array([0.0, 0.0, 0.0]), aabb_scale=4,
                                           render_aabb=bounding_box)
        self.ngp.nerf.training.n_images_for_training = 0
        self.ngp.reload_network_from_file(file_instant_ngp_config)
        self.ngp.shall_train = True
        self.ngp.nerf.training.optimize_extrinsics = True
        self.ngp.nerf.training.depth_loss_type = pyngp.LossType.L2
        self.ngp.frame()

    def __call__(self, input: dict | None) -> None:
        if self.step == self.args.mapping_iterations:
            self.shut_down()
        else:
            if input is not None:
                self.update(input=input)
            if self.is_initialized:
                self.train()

    def update(self, input: dict) -> None:
        color = input["frames_color"].to(self.device).permute(0, 2, 3, 1)
        # sRGB -> linear, exactly as the reference does before handing images to the testbed
        color = torch.where(color > 0.04045, torch.pow((color + 0.055) / 1.055, 2.4), color / 12.92)
        color = torch.cat((color, torch.ones_like(color[..., :1])), dim=3)
        depth = input["frames_depth"].to(self.device).permute(0, 2, 3, 1)
        if "frames_depth_covariance" in input:
            depth_cov = input["frames_depth_covariance"].to(self.device).permute(0, 2, 3, 1)
        else:
            depth_cov = torch.ones_like(depth)
        self.ngp.nerf.training.update_training_images(
            frame_ids=input["keyframe_indices"].contiguous().cpu().numpy().tolist(),
            poses=input["camera_extrinsics"].to(self.device)[:, :3].contiguous(),
            images=color.contiguous(), depths=depth.contiguous(), depths_cov=depth_cov.contiguous(),
            resolution=np.array([self.args.frame_width, self.args.frame_height]),
            principal_point=input["camera_intrinsics"][0, 2:].cpu().numpy(),
            focal_length=input["camera_intrinsics"][0, :2].cpu().numpy(),
            depth_scale=1.0, depth_cov_scale=1.0)
        self.is_initialized = True  # (no torch.cuda.empty_cache(): allocator churn)

    def train(self) -> None:
        self.ngp.frame()
        if step_check(self.step, self.args.mapping_snapshot_iterations):
            self.save_snapshot()
        self.step += 1

    def shut_down(self) -> None:
        self.save_snapshot()
        self.is_shut_down = True

    def save_snapshot(self) -> None:
        self.ngp.save_snapshot(self.args.dir_prediction + f"/snapshots/snapshot{self.step:06d}.msgpack", False)


class InstantNGPRenderer(NeRFRenderer):
    """render_frame(intrinsics, extrinsics) -> (uint8 sRGB colour, z-depth): linear render -> un-premultiply ->
    sRGB transfer -> clip -> *255 + 0.5, through the same testbed calls as the reference."""

    def load_nerf_from_snapshot(self, dir_prediction: str) -> None:
        dir_snapshots = dir_prediction + "/snapshots"
        files = sorted(f for f in os.listdir(dir_snapshots) if ".msgpack" in f) if os.path.exists(dir_snapshots) else []
        if not files:
            raise FileNotFoundError(f"Could not find snapshot in {dir_snapshots}")
        self.load_ngp_from_snapshot(file_snapshot=os.path.join(dir_snapshots, files[-1]))

    def load_ngp_from_snapshot(self, file_snapshot: str) -> None:
        self.ngp = pyngp.Testbed(pyngp.TestbedMode.Nerf, 0)
        self.ngp.load_snapshot(path=file_snapshot)

    def load_nerf_from_mapping_model(self, mapping_model) -> None:
        self.ngp = mapping_model.ngp

    def get_camera_extrinsics(self, frame_index: int) -> np.ndarray:
        m = np.eye(4)
        m[:3] = self.ngp.nerf.training.get_camera_extrinsics(frame_idx=frame_index)
        m = m[[1, 2, 0, 3]]   # NGP row order -> NeRF
        m[0:3, 1:3] *= -1     # NeRF (y up, -z forward) -> standard (y down, z forward)
        return m

    def render_frame(self, camera_intrinsics: dict, camera_extrinsics: np.ndarray) -> tuple:
        color = self.render_frame_color(camera_intrinsics=camera_intrinsics, camera_extrinsics=camera_extrinsics.copy())
        depth = self.render_frame_depth(camera_intrinsics=camera_intrinsics, camera_extrinsics=camera_extrinsics.copy())
        return color, depth

    def render_frame_color(self, camera_intrinsics: dict, camera_extrinsics: np.ndarray) -> np.ndarray:
        self._set_rendering_defaults(camera_intrinsics)
        self._set_camera_extrinsics(camera_extrinsics)
        self.ngp.render_mode = pyngp.Shade
        color = self.ngp.render(width=camera_intrinsics["width"], height=camera_intrinsics["height"], spp=1, linear=True)
        color[..., 0:3] = np.divide(color[..., 0:3], color[..., 3:4], out=np.zeros_like(color[..., 0:3]),
                                    where=color[..., 3:4] != 0)
        lin = color[..., 0:3]
        srgb = np.where(lin > 0.0031308, 1.055 * (np.maximum(lin, 1e-12) ** (1.0 / 2.4)) - 0.055, 12.92 * lin)
        return (np.clip(srgb, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)

    def render_frame_depth(self, camera_intrinsics: dict, camera_extrinsics: np.ndarray) -> np.ndarray:
        self._set_rendering_defaults(camera_intrinsics)
        self._set_camera_extrinsics(camera_extrinsics)
        self.ngp.render_mode = pyngp.Depth
        return self.ngp.render(width=camera_intrinsics["width"], height=camera_intrinsics["height"], spp=1,
                               linear=True)[..., 0]

    def render_mesh(self, file_mesh: str, resolution, lower_bound, upper_bound) -> None:
        """ref: evaluation/nerf_renderer.py:296-300"""
        from .. import pyngp

        bounding_box = pyngp.BoundingBox(lower_bound, upper_bound)
        self.ngp.compute_and_save_marching_cubes_mesh(file_mesh, resolution, bounding_box)

    def _set_rendering_defaults(self, camera_intrinsics: dict) -> None:
        self.ngp.nerf.sharpen = 0.0
        self.ngp.exposure = 0.0
        self.ngp.fov_axis = 0
        self.ngp.fov = (2 * math.atan(0.5 * camera_intrinsics["width"] / camera_intrinsics["fx"])) * 180 / np.pi
        self.ngp.nerf.render_with_lens_distortion = True
        self.ngp.nerf.render_min_transmittance = 1e-4

    def _set_camera_extrinsics(self, camera_extrinsics: np.ndarray) -> None:
        camera_extrinsics = np.array(camera_extrinsics, dtype=np.float64, copy=True)
        camera_extrinsics[0:3, 1:3] *= -1              # standard -> NeRF convention
        self.ngp.set_nerf_camera_matrix(camera_extrinsics[[2, 0, 1]])  # NeRF -> NGP row order


class NeRFSLAMNGPRenderer(InstantNGPRenderer):
    """Same renderer; the reference's subclass only locates the NeRF-SLAM build of pyngp (nerf_renderer.py:322-344)."""
