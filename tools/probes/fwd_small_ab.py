#!/usr/bin/env python3
"""A/B of the small-grid forward forms (NVO_GRID_FWD_SMALL = 1 plain | 3 software-pipelined): run once per value, the
second run compares its outputs bit for bit with what the first one saved, both print their launch times.
Usage: NVO_GRID_FWD_SMALL=1 python tools/probes/fwd_small_ab.py /tmp/ab.pt ; NVO_GRID_FWD_SMALL=3 python ... /tmp/ab.pt"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry  # noqa: E402

entry.build()
import nerf_vo_amd.tinycudann as tcnn  # noqa: E402
from nerf_vo_amd import _lib  # noqa: E402


def pls(b, m, L):
    return float(np.exp((np.log(m) - np.log(b)) / (L - 1)))


def main():
    path = sys.argv[1]
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    torch.manual_seed(0)
    outs = {}
    for label, mx, n in (("prop0", 128, 4096 * 256), ("prop1", 256, 4096 * 96), ("prop0-ragged", 128, 100_003 * 4),
                         ("prop1-small", 256, 2048), ("prop0-render", 128, 32768 * 256)):
        # (the stand-alone Encoding writes sample-major rows; the level-major small-grid forward runs inside
        # NetworkWithInputEncoding, as in the engine's proposal networks)
        enc = tcnn.NetworkWithInputEncoding(3, 1, {"otype": "HashGrid", "n_levels": 5, "n_features_per_level": 2,
                                                   "log2_hashmap_size": 17, "base_resolution": 16, "per_level_scale": pls(16, mx, 5)},
                                            {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                                             "n_neurons": 16, "n_hidden_layers": 1}).to(dev)
        with torch.no_grad():
            enc.params.uniform_(-1, 1)
        # ray-coherent positions as the proposal sampler produces them: S consecutive samples per ray, uniform in
        # disparity, scene-contracted into [0, 1]^3 (random positions make every lane of a gather its own cache line)
        S = 96 if label.startswith("prop1") else 256
        R = (n + S - 1) // S
        o = (torch.rand(R, 1, 3, device=dev) - 0.5) * 1.2
        d = torch.nn.functional.normalize(torch.randn(R, 1, 3, device=dev), dim=-1)
        t = 1.0 / torch.linspace(1.0 / 0.05, 1.0 / 30.0, S, device=dev).view(1, S, 1)
        p = o + d * t
        mag = p.abs().amax(dim=-1, keepdim=True).clamp_min(1e-9)
        p = torch.where(mag > 1, (2 - 1 / mag) * (p / mag), p)  # L-inf scene contraction
        x = ((p + 2) / 4).reshape(-1, 3)[:n].contiguous()
        x[:64] = torch.tensor([0.0, 1.0, 0.5], device=dev)  # domain faces: the dense levels' wrap
        x[64:128] = 1.0
        for it in range(13):
            if it == 3:
                torch.cuda.synchronize()
                lib.nvo_profile_enable(1)
            with torch.no_grad():
                y = enc(x)
        torch.cuda.synchronize()
        if n % 128:  # (the native entry takes whole 128-sample tiles: the module pads, the raw call below does not)
            outs[label] = y.view(torch.int16).cpu() if y.dtype == torch.float16 else y.cpu()
            continue
        # back-to-back launches of the NATIVE forward alone (no per-launch event pair): best of 5 rounds of 40
        mod = enc.native_tcnn_module
        from nerf_vo_amd.tinycudann.modules import _ptr, _stream
        half = enc.params.detach().to(torch.float16)
        outb = torch.empty(n, dtype=torch.float16, device=dev)
        ctx = torch.empty(mod.ctx_bytes(n), dtype=torch.uint8, device=dev)
        mod.set_option("compact_output", 1)
        best = 1e9
        for rnd in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                _lib.check(lib.nvo_fwd(mod.handle, _stream(dev), n, _ptr(x), _ptr(half), _ptr(outb), _ptr(ctx)), "nvo_fwd")
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 40 * 1e3)
        mod.set_option("compact_output", 0)
        print(f"{label:14s} N={n:9d} grid_fwd + mlp_fwd back to back: {best:8.1f} us per pair (best of 5 x 40)")
        need = lib.nvo_profile_summary(None, 0)
        buf = C.create_string_buffer(int(need) + 16)
        lib.nvo_profile_summary(buf, len(buf))
        lib.nvo_profile_enable(0)
        for line in buf.value.decode().strip().splitlines():
            name, cnt, total = line.rsplit(",", 2)
            print(f"{label:14s} N={n:9d} {name:20s} avg {float(total) / int(cnt) * 1e3:8.1f} us  (NVO_GRID_FWD_SMALL={os.environ.get('NVO_GRID_FWD_SMALL', 'default')})")
        outs[label] = y.view(torch.int16).cpu() if y.dtype == torch.float16 else y.cpu()
    if os.path.exists(path):
        ref = torch.load(path)
        for k, v in outs.items():
            same = bool(torch.equal(ref[k], v))
            print(f"{k:14s} bit-identical to the saved run: {same}")
            assert same, k
    else:
        torch.save(outs, path)
        print(f"saved {path}")


if __name__ == "__main__":
    main()
