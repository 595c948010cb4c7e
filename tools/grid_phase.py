#!/usr/bin/env python3
"""Shader-clock shares per phase of the hash-grid kernels inside real training steps, from a library built with
NVO_EXTRA_CXXFLAGS=-DNVO_GRID_PHASE (set the variable for the build AND for this run, or the stamp check rebuilds the
product library).  Usage: NVO_EXTRA_CXXFLAGS=-DNVO_GRID_PHASE python tools/grid_phase.py [--steps 120] [--mlp-dtype bf16]
[--dynamic-loss-scale].  The instrumented launches are slower than the product's (one atomic per phase and workgroup):
read the shares and the per-item / per-workgroup cycle counts, not the totals.  DESIGN.md section 3.6 quotes them."""
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


def table(title, names, vals, per, per_name):
    tot = sum(vals)
    print(f"{title}: {per:.0f} {per_name}; {tot / max(per, 1):.0f} cycles each")
    for n, v in zip(names, vals):
        print(f"    {n:34s} {v / max(per, 1):10.0f} cycles  {100.0 * v / max(tot, 1):5.1f} %")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--skip", type=int, default=20, help="steps before the counters are reset")
    ap.add_argument("--keyframes", type=int, default=48)
    ap.add_argument("--mlp-dtype", default="f16")
    ap.add_argument("--dynamic-loss-scale", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n, H, W, R = a.keyframes, 240, 320, 4096
    torch.manual_seed(0)
    seq = make_sequence(n, H, W, device=dev)
    ds = DynamicDataset(num_frames=n, frame_height=H, frame_width=W, device=dev, use_normals=False)
    ds.update({"keyframe_indices": torch.arange(n), "camera_intrinsics": seq["camera_intrinsics"],
               "camera_extrinsics": opencv_to_opengl(seq["camera_extrinsics"]), "frames_color": seq["frames_color"],
               "frames_depth": seq["frames_depth"]})
    eng = NerfactoEngine(EngineConfig(num_images=n, num_rays=R, mlp_dtype=a.mlp_dtype,
                                      dynamic_loss_scale=a.dynamic_loss_scale), dev)
    raw = _lib.lib()
    if not hasattr(raw, "nvo_debug_grid_phase"):
        raise SystemExit("library was not built with -DNVO_GRID_PHASE (set NVO_EXTRA_CXXFLAGS for this run)")
    out = (C.c_ulonglong * 48)()
    for it in range(a.steps):
        if it == a.skip:
            torch.cuda.synchronize()
            assert raw.nvo_debug_grid_phase(out, 1) == 0
        eng.train_step_graphed(ds)
    torch.cuda.synchronize()
    assert raw.nvo_debug_grid_phase(out, 0) == 0
    v = [int(x) for x in out]
    print(f"steps {a.skip}..{a.steps} of a {n}-keyframe run, {a.mlp_dtype}, loss scale "
          f"{'dynamic (65536)' if a.dynamic_loss_scale else 'static 128'}; cycles = s_memtime ticks")
    table("k_tl_accumulate_p, per item", ["zero + L1 bound + barrier", "record loop (loads, atomics)", "barrier behind slowest wave",
                                          "flush", "end barrier"], v[0:5], v[5] + v[6], f"items ({v[5]} hashed, {v[6]} dense chunks)")
    table("k_tl_scatter_p, per workgroup", ["first loads (x, dy) + tile max", "cell, hashes, rank atomics", "barrier",
                                            "bin scan (wave 0) + barrier", "staging + barrier", "copy-out issue"], v[8:14], v[14],
          "workgroups")
    table("slice-owner items of DENSE levels", ["zero + barrier", "scan", "barrier behind slowest wave", "flush"], v[16:20], v[20], "items")
    table("slice-owner items of HASHED levels", ["zero + barrier", "scan", "barrier behind slowest wave", "flush"], v[24:28], v[28], "items")
    table("k_grid_fwd_small, per workgroup", ["LDS staging + barrier", "sample loop"], v[32:34], v[37], "workgroups")
    table("  sample loop of k_grid_fwd_small (lean form: slot 0 = staging alone, before the first pass's index arithmetic)",
          ["cells, hashes, gather issue", "LDS levels", "gather wait + interpolation + stores"], v[34:37], v[37], "workgroups")



if __name__ == "__main__":
    main()
