#!/usr/bin/env python3
"""Per-phase shader cycles of the colour-head backward's tile loop (wave 0), from a library built with
NVO_EXTRA_CXXFLAGS=-DNVO_MLP_PHASE (set the variable for the build AND for this run, or the stamp check rebuilds the
product library).  Usage: NVO_EXTRA_CXXFLAGS=-DNVO_MLP_PHASE python tools/mlp_phase.py"""
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

NAMES = ["issue next tile's loads", "output layer + forward recompute", "dW_last (transposes + MFMA)", "dZ last hidden",
         "hidden layers (transposes, dW, chain)", "dW0 (transposes + MFMA)", "dX chain", "epilogue (stores, atomics)",
         "cur = nxt (wait for the prefetch)", "loop overhead", "PROLOGUE (weights, first tile), whole kernel",
         "dW FLUSH (LDS reduction + atomics), whole kernel"]


def main():
    dev = torch.device("cuda:0")
    n, H, W, R = 48, 240, 320, 4096
    torch.manual_seed(0)
    seq = make_sequence(n, H, W, device=dev)
    ds = DynamicDataset(num_frames=n, frame_height=H, frame_width=W, device=dev, use_normals=False)
    ds.update({"keyframe_indices": torch.arange(n), "camera_intrinsics": seq["camera_intrinsics"],
               "camera_extrinsics": opencv_to_opengl(seq["camera_extrinsics"]), "frames_color": seq["frames_color"],
               "frames_depth": seq["frames_depth"]})
    # NVO_PHASE_DENSITY_BIAS=12: the state of a trained field (a ray's first sample takes its whole weight, most tiles dead)
    bias = float(os.environ.get("NVO_PHASE_DENSITY_BIAS", "-1"))
    eng = NerfactoEngine(EngineConfig(num_images=n, num_rays=R, density_bias=bias, dynamic_loss_scale=bias < 0), dev)
    for _ in range(60):
        eng.train_step_graphed(ds)
    torch.cuda.synchronize()
    raw = C.CDLL(str(_lib.LIB_PATH)) if hasattr(_lib, "LIB_PATH") else _lib.lib()
    out = (C.c_ulonglong * 16)()
    rc = raw.nvo_debug_mlp_phase(out)
    assert rc == 0, "library was not built with -DNVO_MLP_PHASE"
    if "tile_live" in eng._workspace(R, True):
        live = eng._workspace(R, True)["tile_live"]
        print(f"tiles with an rgb gradient {float((live & 1).bool().float().mean()):.3f}, with any {float((live != 0).float().mean()):.3f}")
    tot = sum(out[k] for k in range(10))
    tiles = R * 48 // 16 // 1024
    print(f"wave 0: {tot} cycles in the tile loop, {tiles} tiles -> {tot / tiles:.0f} cycles per tile")
    for k in (9, 0, 1, 2, 3, 4, 5, 6, 7, 8):
        print(f"  {NAMES[k]:42s} {out[k] / tiles:9.0f} cycles/tile  {100.0 * out[k] / tot:5.1f} %")
    for k in (10, 11):
        print(f"  {NAMES[k]:50s} {out[k]:9d} cycles")


if __name__ == "__main__":
    main()
