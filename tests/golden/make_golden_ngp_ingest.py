#!/usr/bin/env python3
"""Golden vectors for the ingest of the occupancy-grid back-end (ingest half), produced BY
THE REFERENCE's own method.

/root/reference/nerf_vo/mapping/instant_ngp.py cannot be imported (``import pyngp``), but ``InstantNGP.update`` (:61-102)
-- sRGB -> linear, alpha channel, NCHW -> NHWC, default depth covariance, the argument list of
``update_training_images`` -- is plain torch / numpy.  Its definition is parsed out of the reference file and executed AT
GENERATION TIME (nothing is copied into the repository), bound to an ``object.__new__`` instance whose ``self.ngp`` is a
recorder: what the testbed WOULD have received is what is stored.  Two packets: without and with a depth covariance.
Writes tests/golden/ngp_ingest_golden.npz.   python tests/golden/make_golden_ngp_ingest.py
"""
import argparse
import ast
import os
import types

import numpy as np
import torch

REF = "/root/reference/nerf_vo/mapping/instant_ngp.py"
H, W = 6, 8


def reference_update():
    tree = ast.parse(open(REF).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "InstantNGP")
    cls.body = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "update"]
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), os.path.basename(REF), "exec"), ns)
    return ns["InstantNGP"]


def build():
    cls = reference_update()
    out = {"height": np.array(H), "width": np.array(W)}
    g = torch.Generator().manual_seed(31)
    for i, (k, with_cov) in enumerate(((3, False), (2, True))):
        received = {}
        self = object.__new__(cls)
        self.device = torch.device("cpu")
        self.args = argparse.Namespace(frame_width=W, frame_height=H)
        self.is_initialized = False
        self.ngp = types.SimpleNamespace(nerf=types.SimpleNamespace(training=types.SimpleNamespace(
            update_training_images=lambda **kw: received.update(kw))))
        packet = {"keyframe_indices": torch.tensor([4, 1, 7][:k], dtype=torch.long),
                  "camera_intrinsics": torch.rand(k, 4, generator=g) * 100 + 200,
                  "camera_extrinsics": torch.randn(k, 4, 4, generator=g),
                  # colours on both sides of the sRGB knee (0.04045) and at the ends of the range
                  "frames_color": torch.cat([torch.rand(k, 3, H, W // 2, generator=g) * 0.08,
                                             torch.rand(k, 3, H, W - W // 2, generator=g)], dim=3),
                  "frames_depth": torch.rand(k, 1, H, W, generator=g) * 5}
        packet["frames_color"][0, :, 0, 0] = torch.tensor([0.0, 0.04045, 1.0])
        if with_cov:
            packet["frames_depth_covariance"] = torch.rand(k, 1, H, W, generator=g)
        for name, v in packet.items():
            out[f"p{i}_in_{name}"] = v.numpy().copy()
        self.update(input=packet)
        assert self.is_initialized is True
        out[f"p{i}_kwargs"] = np.array(sorted(received))
        out[f"p{i}_frame_ids"] = np.array(received["frame_ids"])
        for name in ("poses", "images", "depths", "depths_cov"):  # lists of per-frame numpy arrays
            assert isinstance(received[name], list) and len(received[name]) == k
            out[f"p{i}_{name}"] = np.stack(received[name])
        for name in ("resolution", "principal_point", "focal_length"):
            out[f"p{i}_{name}"] = np.asarray(received[name])
        out[f"p{i}_depth_scale"] = np.array(received["depth_scale"])
        out[f"p{i}_depth_cov_scale"] = np.array(received["depth_cov_scale"])
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ngp_ingest_golden.npz")
    np.savez_compressed(path, **build())
    print("wrote", path, os.path.getsize(path), "bytes")
