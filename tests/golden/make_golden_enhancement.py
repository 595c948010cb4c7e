#!/usr/bin/env python3
"""Golden vectors for the keyframe depth alignment, produced BY THE REFERENCE:
imports /root/reference/nerf_vo/enhancement/enhancement_module.py (pure torch, runs on CPU) and executes
EnhancementModule.step() with a stub monocular estimator.  Only inputs and the reference's outputs are
stored (tests/golden/enhancement_golden.npz); this script needs /root/reference and is not run on the GPU box.

    python tests/golden/make_golden_enhancement.py
"""
import argparse
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from nerf_vo.enhancement.enhancement_module import EnhancementModule  # noqa: E402


class _StubEstimator:
    """Stands in for OmnidataEstimator: returns the prepared monocular depth / normals, consumes no RNG."""
    is_initialized = True

    def __init__(self, depth, normal):
        self.depth, self.normal = depth, normal

    def __call__(self, frames_color):
        return self.depth.clone(), None if self.normal is None else self.normal.clone()


def run_reference(frames_color_u8, depth, normal, patches, seed, removal_window=28, mapping_module="nerfstudio"):
    K = patches.shape[0]
    mod = object.__new__(EnhancementModule)
    mod.name = "depth-normal" if normal is not None else "depth"
    mod.args = argparse.Namespace(tracking_module="dpvo", removal_window=removal_window, mapping_module=mapping_module,
                                  num_keyframes=64, frame_height=depth.shape[2], frame_width=depth.shape[3])
    mod.device = torch.device("cpu")
    mod.method = _StubEstimator(depth, normal)
    mod.step_counter = 0
    mod.shutdown = False
    mod.shared_variables = {"status_lock": threading.Lock(), "status": {}}
    from collections import deque
    mod.buffer_camera_intrinsics = deque(maxlen=removal_window - 2)
    mod.buffer_frames_color = deque(maxlen=removal_window - 2)
    mod.buffer_frames_depth = deque(maxlen=removal_window - 2)
    inp = {"keyframe_indices": torch.arange(K), "camera_intrinsics": torch.rand(K, 4),
           "camera_extrinsics": torch.eye(4).repeat(K, 1, 1) + 0.01 * torch.arange(16.0).reshape(4, 4),
           "frames_color": frames_color_u8.float(), "dpvo_patches": patches.clone(), "last_frame": False}
    extr_in = inp["camera_extrinsics"].clone()
    torch.manual_seed(seed)
    out, skip = mod.step(inp)
    assert not skip
    torch.manual_seed(seed)
    noise = torch.rand((K, patches.shape[1], 1, 1), dtype=torch.float32)  # the stream dpvo_remove_outliers consumed
    return out, noise, extr_in


def build():
    g = torch.Generator().manual_seed(2024)
    out = {}
    cases = {"a": dict(K=5, M=96, H=48, W=64, normal=True), "b": dict(K=3, M=96, H=40, W=56, normal=False),
             "fallback": dict(K=2, M=90, H=32, W=48, normal=False)}  # M=90: survivors != 75 -> the `except` path
    for name, c in cases.items():
        K, M, H, W = c["K"], c["M"], c["H"], c["W"]
        depth = torch.rand(K, 1, H, W, generator=g) * 2.0 + 0.3
        normal = torch.rand(K, 3, H, W, generator=g) if c["normal"] else None
        color = torch.randint(0, 256, (K, 3, H, W), generator=g, dtype=torch.uint8)
        patches = torch.zeros(K, M, 3, 3, 3)
        patches[:, :, 0] = (torch.rand(K, M, 1, 1, generator=g) * (W / 4 - 1)).expand(K, M, 3, 3)
        patches[:, :, 1] = (torch.rand(K, M, 1, 1, generator=g) * (H / 4 - 1)).expand(K, M, 3, 3)
        patches[:, :, 2] = (0.15 + torch.rand(K, M, 1, 1, generator=g) * 2.5).expand(K, M, 3, 3)
        patches[0, 0, 2] = 0.0  # a degenerate inverse depth: 1/0 -> inf -> clipped to 5
        res, noise, extr_in = run_reference(color, depth, normal, patches, seed=7 + len(name))
        out[f"{name}_frames_color_u8"] = color.numpy()
        out[f"{name}_mono_depth"] = depth.numpy()
        out[f"{name}_patches"] = patches.numpy()
        out[f"{name}_noise"] = noise.numpy()
        out[f"{name}_extrinsics_in"] = extr_in.numpy()
        out[f"{name}_ref_frames_depth"] = res["frames_depth"].numpy()
        out[f"{name}_ref_frames_color"] = res["frames_color"].numpy()
        out[f"{name}_ref_extrinsics"] = res["camera_extrinsics"].numpy()
        if normal is not None:
            out[f"{name}_mono_normal"] = normal.numpy()
            out[f"{name}_ref_frames_normal"] = res["frames_normal"].numpy()
    return out


if __name__ == "__main__":
    data = build()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "enhancement_golden.npz")
    np.savez_compressed(path, **data)
    print(f"wrote {path}: {len(data)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")
