#!/usr/bin/env python3
"""Kernel-level timing on the GPU box (HIP-event profiler of the library): hash-grid forward /
backward variants at the BASELINE batch sizes.  Usage: python tools/kernel_bench.py [--iters 20]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

entry.build()
import nerf_vo_amd.tinycudann as tcnn  # noqa: E402
from nerf_vo_amd import _lib  # noqa: E402


def pls(b, m, L):
    return float(np.exp((np.log(m) - np.log(b)) / (L - 1)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--modes", type=int, nargs="+", default=[0, 1, 2, 3])
    ap.add_argument("--tiles", type=int, nargs="+", default=[512], help="mode 3: samples per count/scatter tile")
    ap.add_argument("--masks", type=str, nargs="+", default=["12"], help="mode 3: owner_max_slices values")
    ap.add_argument("--cases", type=int, nargs="+", default=[0, 1, 2])
    ap.add_argument("--acc-bits", type=int, default=64, help="accumulators of the slice-owner items (32 | 64)")
    ap.add_argument("--random-x", action="store_true", help="uniform random positions instead of ray-coherent ones")
    ap.add_argument("--show-fwd", action="store_true", help="also print the forward gather's time")
    ap.add_argument("--runs", type=int, default=1, help="run-merging scan of the slice-owner items (grid_bwd_runs)")
    ap.add_argument("--phase", action="store_true",
                    help="library built with NVO_EXTRA_CXXFLAGS=-DNVO_GRID_PHASE: shader clocks per phase of the slice-owner items")
    ap.add_argument("--dead", type=float, default=0.0, help="share of the samples whose dL/dy is exactly zero (ray-coherent runs)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    cases = [("main L16 T19", dict(n_levels=16, log2_hashmap_size=19, base_resolution=16, per_level_scale=pls(16, 2048, 16)), 4096 * 48),
             ("prop0 L5 T17", dict(n_levels=5, log2_hashmap_size=17, base_resolution=16, per_level_scale=pls(16, 128, 5)), 4096 * 256),
             ("prop1 L5 T17", dict(n_levels=5, log2_hashmap_size=17, base_resolution=16, per_level_scale=pls(16, 256, 5)), 4096 * 96)]
    for label, cfg, n in [cases[c] for c in args.cases]:
        enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_features_per_level": 2, **cfg}).to(dev)
        with torch.no_grad():
            enc.params.uniform_(-1, 1)
        # ray-coherent positions (48 consecutive samples along a ray), like the training step
        R = (n + 47) // 48
        o = (torch.rand(R, 1, 3, device=dev) - 0.5) * 0.5 + 0.5
        d = torch.nn.functional.normalize(torch.randn(R, 1, 3, device=dev), dim=-1)
        t = torch.linspace(0, 0.4, 48, device=dev).view(1, 48, 1)
        x = (o + d * t).clamp(0.001, 0.999).reshape(-1, 3)[:n].contiguous().requires_grad_(False)
        if args.random_x:
            x = torch.rand(n, 3, device=dev)
        dy = torch.randn(n, enc.n_output_dims, device=dev)
        if args.dead > 0:  # dead samples come in runs along the rays (what the proposal losses produce)
            runs = (torch.rand((n + 15) // 16, device=dev) < args.dead).repeat_interleave(16)[:n]
            dy[runs] = 0
        variants = []
        for mode in args.modes:
            if mode == 3:
                variants += [(3, t, int(m, 0), 1, 32) for t in args.tiles for m in args.masks]
            else:
                variants.append((mode, 0, 0xFFFFFFFF, 0, 64))
        enc.native_tcnn_module.set_option("grid_acc_bits", args.acc_bits)
        enc.native_tcnn_module.set_option("grid_bwd_runs", args.runs)
        enc.native_tcnn_module.set_option("grid_bwd_batch", n)
        for mode, tile, mask, layout, sab in variants:
            enc.native_tcnn_module.set_option("grid_bwd_mode", mode)
            if mode == 3:
                enc.native_tcnn_module.set_option("grid_stream_tile", tile)
                enc.native_tcnn_module.set_option("grid_stream_owner_slices", mask)
            for it in range(args.iters + 3):
                if it == 3:
                    torch.cuda.synchronize()
                    lib.nvo_profile_enable(1)
                    if args.phase:
                        ph = (C.c_ulonglong * 48)()
                        assert lib.nvo_debug_grid_phase(ph, 1) == 0
                enc.params.grad = None
                y = enc(x)
                (y.float() * dy).sum().backward()
            torch.cuda.synchronize()
            need = lib.nvo_profile_summary(None, 0)
            buf = C.create_string_buffer(int(need) + 16)
            lib.nvo_profile_summary(buf, len(buf))
            lib.nvo_profile_enable(0)
            for line in buf.value.decode().strip().splitlines():
                name, cnt, total = line.rsplit(",", 2)
                if name.startswith("grid_fwd") and not args.show_fwd:
                    continue
                tag = f"mode={mode}" + (f" tile={tile} owner<={mask} layout={layout} acc={sab}" if mode == 3 else "")
                print(f"{label:14s} N={n:8d} {tag:32s} {name:24s} avg {float(total) / int(cnt) * 1e3:9.1f} us")
            if args.phase:
                assert lib.nvo_debug_grid_phase(ph, 0) == 0
                v = [int(q) for q in ph]
                for kind, o in (("dense", 16), ("hashed", 24)):
                    k = max(v[o + 4], 1)
                    print(f"{label:14s}   {kind:6s} items/launch {v[o + 4] / args.iters:6.0f}  cycles/item: zero {v[o] / k:8.0f} "
                          f"scan {v[o + 1] / k:8.0f} barrier {v[o + 2] / k:8.0f} flush {v[o + 3] / k:8.0f}")
                print(f"{label:14s}   scan cycles per item by level: " + "  ".join(
                    f"L{l}: {v[38 + l] / max(v[43 + l], 1):.0f} ({v[43 + l] / args.iters:.0f} items)" for l in range(5)))


if __name__ == "__main__":
    main()
