"""nerfstudio-shaped ``Cameras`` / ``RayBundle`` (the subset the reference uses:
/root/reference/nerf_vo/mapping/nerfstudio_utils.py:90-107, /root/reference/evaluation/
nerf_renderer.py:136-165, /root/reference/nerf_vo/mapping/nerfstudio.py:134-135).  Ray generation is
the HIP kernel nvo_raygen; this file only holds tensors."""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass, field

import torch

from .. import _lib
from ..tinycudann.modules import _ptr, _stream


class CameraType(enum.Enum):
    PERSPECTIVE = 1


@dataclass
class RayBundle:
    origins: torch.Tensor
    directions: torch.Tensor
    pixel_area: torch.Tensor
    camera_indices: torch.Tensor | None = None
    nears: torch.Tensor | None = None
    fars: torch.Tensor | None = None
    metadata: dict = field(default_factory=dict)

    @property
    def shape(self):
        return self.origins.shape[:-1]

    def __len__(self):
        return int(self.origins.shape[:-1].numel())


def _as_tensor(v, n: int | None = None) -> torch.Tensor:
    t = v if isinstance(v, torch.Tensor) else torch.tensor(v, dtype=torch.float32)
    t = t.to(torch.float32)
    if t.ndim == 0:
        t = t[None]
    return t


class Cameras:
    """Pinhole cameras.  fx/fy/cx/cy and camera_to_worlds may be VIEWS of caller-owned buffers (the
    reference builds its Cameras once over the dataset's shared buffers and relies on in-place
    updates being visible -- nerfstudio_utils.py:90-107); nothing is copied here unless .to() has to
    move devices."""

    def __init__(self, camera_to_worlds, fx, fy, cx, cy, width, height, distortion_params=None,
                 camera_type=CameraType.PERSPECTIVE):
        self.camera_to_worlds = camera_to_worlds if camera_to_worlds.ndim == 3 else camera_to_worlds[None]
        n = self.camera_to_worlds.shape[0]
        self.fx, self.fy, self.cx, self.cy = (_as_tensor(v, n) for v in (fx, fy, cx, cy))
        self.width, self.height = int(width), int(height)
        self.distortion_params = distortion_params
        self.camera_type = camera_type

    @property
    def device(self):
        return self.camera_to_worlds.device

    def __len__(self):
        return self.camera_to_worlds.shape[0]

    def to(self, device):
        device = torch.device(device)
        if all(t.device == device for t in (self.camera_to_worlds, self.fx, self.fy, self.cx, self.cy)):
            return self  # (views of caller-owned buffers stay views)
        return Cameras(self.camera_to_worlds.to(device), self.fx.to(device), self.fy.to(device), self.cx.to(device),
                       self.cy.to(device), self.width, self.height, self.distortion_params, self.camera_type)

    def intrinsics_matrix(self) -> torch.Tensor:
        n = len(self)
        cols = [v.expand(n) if v.numel() == 1 else v for v in (self.fx, self.fy, self.cx, self.cy)]
        return torch.stack(cols, dim=1).contiguous()

    def generate_rays(self, camera_indices, coords=None, keep_shape: bool = True, corrections=None) -> RayBundle:
        """camera_indices: int -> all pixels of that camera ([H,W] bundle when keep_shape);
        or an int64 tensor [R,3] of (camera, y, x)."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("Cameras.generate_rays runs the HIP ray generator: move the cameras to the GPU")
        if isinstance(camera_indices, int):
            ys, xs = torch.meshgrid(torch.arange(self.height, device=dev), torch.arange(self.width, device=dev),
                                    indexing="ij")
            idx = torch.stack([torch.full_like(ys, camera_indices), ys, xs], dim=-1).reshape(-1, 3).contiguous()
            shape = (self.height, self.width) if keep_shape else (self.height * self.width,)
        else:
            idx = camera_indices.to(dev, torch.int64).reshape(-1, 3).contiguous()
            shape = (idx.shape[0],)
        R = idx.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        origins, directions = torch.empty(R, 3, **f32), torch.empty(R, 3, **f32)
        dnorm, area = torch.empty(R, **f32), torch.empty(R, **f32)
        cam = torch.empty(R, dtype=torch.int32, device=dev)
        intr = self.intrinsics_matrix()
        c2w = self.camera_to_worlds[:, :3, :4].contiguous()
        if intr.device != dev or (corrections is not None and corrections.device != dev):
            # (a host pointer handed to the kernel is a GPU memory fault, not an exception)
            raise RuntimeError("Cameras.generate_rays: intrinsics / corrections are not on the cameras' device; call .to(device)")
        _lib.check(_lib.lib().nvo_raygen(_stream(dev), R, _ptr(idx), _ptr(intr), _ptr(c2w), _ptr(corrections),
                                         _ptr(origins), _ptr(directions), _ptr(dnorm), _ptr(area), _ptr(cam)),
                   "nvo_raygen")
        return RayBundle(
            origins=origins.view(*shape, 3), directions=directions.view(*shape, 3), pixel_area=area.view(*shape, 1),
            camera_indices=cam.view(*shape, 1), metadata={"directions_norm": dnorm.view(*shape, 1)})

    def to_json(self, camera_idx: int, image=None, max_size=None) -> dict:
        flat = {"type": "PinholeCamera", "cx": float(self._at(self.cx, camera_idx)),
                "cy": float(self._at(self.cy, camera_idx)), "fx": float(self._at(self.fx, camera_idx)),
                "fy": float(self._at(self.fy, camera_idx)),
                "camera_to_world": self.camera_to_worlds[camera_idx].tolist(), "camera_index": camera_idx,
                "times": 0}
        return flat

    @staticmethod
    def _at(t: torch.Tensor, i: int):
        return t[0] if t.numel() == 1 else t[i]
