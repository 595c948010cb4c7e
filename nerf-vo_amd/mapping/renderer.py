"""Counterpart of the reference's ``NeRFRenderer`` / ``NerfstudioRenderer``
(/root/reference/evaluation/nerf_renderer.py:35-168) and of its PSNR definition
(/root/reference/evaluation/evaluation_utils.py:289-318), over the native engine."""
from __future__ import annotations

import math
from abc import abstractmethod

import numpy as np
import torch

from .cameras import Cameras, CameraType
from .model import multiply


class NeRFRenderer:
    def __init__(self, mapping_model=None, dir_prediction: str | None = None) -> None:
        if mapping_model is None:
            self.load_nerf_from_snapshot(dir_prediction=dir_prediction)
        else:
            self.load_nerf_from_mapping_model(mapping_model=mapping_model)

    @abstractmethod
    def load_nerf_from_snapshot(self, dir_prediction: str) -> None:
        raise NotImplementedError

    @abstractmethod
    def load_nerf_from_mapping_model(self, mapping_model) -> None:
        raise NotImplementedError

    @abstractmethod
    def get_camera_extrinsics(self, frame_index: int) -> np.ndarray:
        raise NotImplementedError

    @abstractmethod
    def render_frame(self, camera_intrinsics: dict, camera_extrinsics: np.ndarray) -> tuple:
        raise NotImplementedError

    def render_frame_color(self, camera_intrinsics: dict, camera_extrinsics: np.ndarray) -> np.ndarray:
        return self.render_frame(camera_intrinsics, camera_extrinsics)[0]

    def render_frame_depth(self, camera_intrinsics: dict, camera_extrinsics: np.ndarray) -> np.ndarray:
        return self.render_frame(camera_intrinsics, camera_extrinsics)[1]

    def render_frame_color_from_training_frame(self, camera_intrinsics: dict, frame_index: int) -> np.ndarray:
        return self.render_frame_color(camera_intrinsics, self.get_camera_extrinsics(frame_index))

    def render_frame_depth_from_training_frame(self, camera_intrinsics: dict, frame_index: int) -> np.ndarray:
        return self.render_frame_depth(camera_intrinsics, self.get_camera_extrinsics(frame_index))


class NerfstudioRenderer(NeRFRenderer):
    def load_nerf_from_snapshot(self, dir_prediction: str) -> None:
        """Offline reload (reference: config.yml + eval_load_checkpoint + DynamicDataset(dir_prediction),
        /root/reference/evaluation/nerf_renderer.py:94-107,211-218): the newest config.yml under
        <dir_prediction>/nerfstudio, the newest checkpoint it points at, dataset.pt and the exported
        training poses."""
        import glob
        import json

        import yaml

        yaml_files = sorted(glob.glob(dir_prediction + "/nerfstudio/**/config.yml", recursive=True))
        if not yaml_files:
            raise FileNotFoundError(f"could not find config.yml under {dir_prediction}/nerfstudio")
        config = yaml.load(open(yaml_files[-1]).read(), Loader=yaml.Loader)
        config.load_dir = config.get_checkpoint_dir()
        config.pipeline.datamanager.dir_prediction = dir_prediction
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        trainer = config.setup(device=device)
        trainer.setup(test_mode="test")
        self.pipeline = trainer.pipeline
        self.pipeline.eval()
        with open(dir_prediction + "/matrices/matrices_origin2frame_training.json") as file:
            self.matrices_origin2frame_training = np.array(json.load(file))

    def load_nerf_from_mapping_model(self, mapping_model) -> None:
        self.pipeline = mapping_model.trainer.pipeline
        ds = self.pipeline.datamanager.train_dataset
        n = ds.num_active_frames
        self.matrices_origin2frame_training = np.tile(np.eye(4), (n, 1, 1))
        corr = self.pipeline.model.camera_optimizer(torch.arange(n).to(self.pipeline.datamanager.device))
        self.matrices_origin2frame_training[:, :3] = multiply(
            corr, ds.cameras.camera_to_worlds[:n].to(self.pipeline.datamanager.device)).detach().cpu().numpy()
        self.pipeline.eval()

    def get_camera_extrinsics(self, frame_index: int) -> np.ndarray:
        m = self.matrices_origin2frame_training[frame_index].copy()
        m[0:3, 1:3] *= -1  # OpenGL (y up, -z forward) -> standard (y down, z forward)
        return m

    def render_frame(self, camera_intrinsics: dict, camera_extrinsics: np.ndarray) -> tuple:
        camera_extrinsics = np.array(camera_extrinsics, dtype=np.float64, copy=True)
        camera_extrinsics[0:3, 1:3] *= -1  # standard -> OpenGL
        cameras = Cameras(
            fx=camera_intrinsics["fx"], fy=camera_intrinsics["fy"], cx=camera_intrinsics["cx"],
            cy=camera_intrinsics["cy"], height=camera_intrinsics["height"], width=camera_intrinsics["width"],
            camera_to_worlds=torch.tensor(camera_extrinsics, dtype=torch.float32).unsqueeze(0)[:, :3],
            camera_type=CameraType.PERSPECTIVE).to(self.pipeline.device)
        bundle = cameras.generate_rays(camera_indices=0, keep_shape=True)
        with torch.no_grad():
            outputs = self.pipeline.model.get_outputs_for_camera_ray_bundle(bundle)
        color = (outputs["rgb"].cpu().numpy() * 255).astype(np.uint8)  # truncation, like the reference
        depth = (outputs["depth"] / bundle.metadata["directions_norm"]).cpu().numpy()[..., 0]  # z-depth
        return color, depth


def calculate_psnr_reference(image1: np.ndarray, image2: np.ndarray) -> float:
    """Bit-faithful to the reference: the subtraction and the square are evaluated in uint8 and wrap
    modulo 256 (evaluation_utils.py:300-305); per channel, then averaged (:310-318)."""
    assert image1.dtype == np.uint8 and image2.dtype == np.uint8
    vals = []
    for c in range(3):
        with np.errstate(over="ignore"):
            mse = np.mean((image1[..., c] - image2[..., c]) ** 2)
        vals.append(float("inf") if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse)))
    return sum(vals) / 3.0


def calculate_psnr_float(image1: np.ndarray, image2: np.ndarray) -> float:
    """Conventional PSNR (float MSE) on the same images, per channel then averaged."""
    a, b = image1.astype(np.float64), image2.astype(np.float64)
    vals = []
    for c in range(3):
        mse = np.mean((a[..., c] - b[..., c]) ** 2)
        vals.append(float("inf") if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse)))
    return sum(vals) / 3.0
