"""Keyframe enhancement step that feeds the mapper: the MI355X counterpart of
``EnhancementModule.step`` for ``tracking_module='dpvo'`` (/root/reference/nerf_vo/enhancement/
enhancement_module.py:41-129).  Same input / output dictionaries; the per-keyframe depth alignment against
DPVO's sparse patches (:61-99, dpvo_remove_outliers :131-146) runs as HIP kernels on the resident tensors
(``nvo_depth_align``) instead of ~25 torch ops with a host-visible exception path.

The monocular estimator itself (Omnidata, ``ref:nerf_vo/enhancement/omnidata_estimator.py``) is outside the
hot path: any callable ``method(frames_color=[n,3,H,W] in [0,1]) -> (frames_depth [n,1,H,W], frames_normal
[n,3,H,W] | None)`` is accepted, exactly the interface the reference calls at :51-52.
"""
from __future__ import annotations

import ctypes as C
from collections import deque

import torch

from .. import _lib


def align_depth_to_patches(frames_depth: torch.Tensor, dpvo_patches: torch.Tensor,
                           noise: torch.Tensor | None = None) -> torch.Tensor:
    """frames_depth [K,1,H,W] (monocular), dpvo_patches [K,M,3,P,P] (x/4, y/4, inverse depth) -> aligned depth
    [K,1,H,W] = clip(depth * scale_k + shift_k, 0, 5).  ``noise`` [K,M] in [0,1): the reference's tie-breaking
    ``torch.rand`` (drawn here when omitted)."""
    if frames_depth.device.type != "cuda":
        raise RuntimeError("align_depth_to_patches needs MI355X-resident tensors; there is no CPU fallback")
    K, M, _, P, _ = dpvo_patches.shape
    H, W = frames_depth.shape[-2:]
    dev = frames_depth.device
    if noise is None:
        noise = torch.rand((K, M), dtype=torch.float32, device=dev)
    depth = frames_depth.reshape(K, H * W).to(torch.float32).contiguous()
    patches = dpvo_patches.to(torch.float32).contiguous()
    noise = noise.reshape(K, M).to(torch.float32).contiguous()
    out = torch.empty_like(depth)
    lib = _lib.lib()
    scratch = torch.empty(int(lib.nvo_depth_align_scratch_bytes(K, M)), dtype=torch.uint8, device=dev)
    args = _lib.DepthAlignArgs(K=K, M=M, P=P, H=H, W=W, patches=patches.data_ptr(), noise=noise.data_ptr(),
                               frames_depth=depth.data_ptr(), out_depth=out.data_ptr(), scratch=scratch.data_ptr(),
                               scale_shift_out=None)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.nvo_depth_align(stream, C.byref(args)), "nvo_depth_align")
    return out.view(K, 1, H, W)


class DpvoDepthEnhancement:
    """Mirror of ``EnhancementModule`` for DPVO tracking: ``step(input) -> (output, skip)`` with the reference's
    dictionary schema (keyframe_indices, camera_intrinsics, camera_extrinsics, frames_color uint8-range,
    dpvo_patches, last_frame) -> adds frames_depth (+ frames_normal), scales colours to [0,1] and flips the pose
    axes for the nerfstudio mapper (:113-114)."""

    def __init__(self, method, removal_window: int = 28, mapping_module: str = "nerfstudio",
                 tracking_module: str = "dpvo"):
        # method=None + tracking_module='droid-slam' is the reference's enhancement 'none' branch (:105-112):
        # DROID-SLAM's dense inverse depth and its covariance are passed through
        self.method = method
        self.mapping_module = mapping_module
        self.tracking_module = tracking_module
        n = removal_window - 2  # ref: enhancement_module.py:32-37
        self.buffer_camera_intrinsics = deque(maxlen=n)
        self.buffer_frames_color = deque(maxlen=n)
        self.buffer_frames_depth = deque(maxlen=n)
        self.window = n
        self.shutdown = False

    def step(self, input: dict | None):
        if input is None:
            return None, True
        out = dict(input)
        out["keyframe_indices"] = input["keyframe_indices"].clone()
        out["camera_intrinsics"] = input["camera_intrinsics"].clone()
        out["camera_extrinsics"] = input["camera_extrinsics"].clone()
        out["frames_color"] = input["frames_color"] / 255.0
        if self.method is None:
            if self.tracking_module != "droid-slam":
                raise NotImplementedError
            out["frames_depth"] = 1 / out.pop("droid_slam_inverse_depth")[:, None, :, :]
            out["frames_depth_covariance"] = out.pop("droid_slam_depth_covariance").clone()[:, None, :, :]
            if self.mapping_module == "nerfstudio":
                out["camera_extrinsics"][:, :3, 1:3] *= -1
            if out.get("last_frame", False):
                self.shutdown = True
            return out, False
        if self.tracking_module != "dpvo":
            raise NotImplementedError
        frames_depth, frames_normal = self.method(frames_color=out["frames_color"].clone())
        if frames_depth is None:
            raise NotImplementedError
        self.buffer_camera_intrinsics.extend(out["camera_intrinsics"])
        self.buffer_frames_color.extend(out["frames_color"])
        self.buffer_frames_depth.extend(frames_depth)
        if frames_depth.shape[0] <= self.window:  # DPVO hands back patches for its whole sliding window
            frames_depth = torch.stack(list(self.buffer_frames_depth))
        patches = out.pop("dpvo_patches")
        out["frames_depth"] = align_depth_to_patches(frames_depth, patches)
        if frames_normal is not None:
            out["frames_normal"] = torch.nn.functional.normalize(frames_normal * 2.0 - 1.0, p=2, dim=1)
        if self.mapping_module == "nerfstudio":
            out["camera_extrinsics"][:, :3, 1:3] *= -1
        if out.get("last_frame", False):
            self.shutdown = True
        return out, False
