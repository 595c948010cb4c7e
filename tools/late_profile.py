#!/usr/bin/env python3
"""Per-kernel times of the mapping step at a LATER training state than bench.py's window: trains N steps on a small
synthetic sequence (graph replay), then profiles 60 eager steps.  Usage: python tools/late_profile.py [--train 1500]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

entry.build()
from nerf_vo_amd import _lib  # noqa: E402
from nerf_vo_amd.engine import EngineConfig, NerfactoEngine  # noqa: E402
from nerf_vo_amd.mapping.dataset import DynamicDataset, opencv_to_opengl  # noqa: E402
from nerf_vo_amd.synthetic import make_sequence  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", type=int, default=1500)
    ap.add_argument("--compact", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n, H, W, R = 48, 240, 320, 4096
    torch.manual_seed(0)
    seq = make_sequence(n, H, W, device=dev)
    ds = DynamicDataset(num_frames=n, frame_height=H, frame_width=W, device=dev, use_normals=False)
    ds.update({"keyframe_indices": torch.arange(n), "camera_intrinsics": seq["camera_intrinsics"],
               "camera_extrinsics": opencv_to_opengl(seq["camera_extrinsics"]), "frames_color": seq["frames_color"],
               "frames_depth": seq["frames_depth"]})
    eng = NerfactoEngine(EngineConfig(num_images=n, num_rays=R), dev)
    for _ in range(a.train):
        eng.train_step_graphed(ds)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(200):
        eng.train_step_graphed(ds)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 200 * 1e3
    lib = _lib.lib()
    c2w = ds.camera_extrinsics[:, :3, :4].contiguous()
    scale = torch.tensor([n, H, W], device=dev)
    lib.nvo_profile_enable(1)
    for _ in range(60):
        idx = torch.floor(torch.rand(R, 3, device=dev) * scale).long()
        eng.train_step(idx, ds.camera_intrinsics, c2w, ds.frames_color, ds.frames_depth)
    torch.cuda.synchronize()
    need = lib.nvo_profile_summary(None, 0)
    buf = C.create_string_buffer(int(need) + 16)
    lib.nvo_profile_summary(buf, len(buf))
    lib.nvo_profile_enable(0)
    print(f"compact={a.compact}: {ms:.3f} ms/step over steps {a.train}..{a.train + 200} (graph replay)")
    rows = []
    for line in buf.value.decode().strip().splitlines():
        name, cnt, total = line.rsplit(",", 2)
        rows.append((name, int(cnt), float(total)))
    for name, cnt, total in sorted(rows, key=lambda r: -r[2])[:14]:
        print(f"  {name:28s} launches {cnt:4d} avg {total / cnt * 1e3:8.1f} us")


if __name__ == "__main__":
    main()
