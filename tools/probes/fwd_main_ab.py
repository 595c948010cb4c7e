#!/usr/bin/env python3
"""A/B of the main grid's forward forms (module option grid_fwd_small_form: 0 first kernel | 4 instruction-lean) inside
ONE process: bit-identity and launch times at the training batch (196 608 samples) and an inference chunk (32 768 x 48),
ray-coherent positions.  Usage: python tools/probes/fwd_main_ab.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry  # noqa: E402

entry.build()
import nerf_vo_amd.tinycudann as tcnn  # noqa: E402
from nerf_vo_amd import _lib  # noqa: E402
from nerf_vo_amd.tinycudann.modules import _ptr, _stream  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    torch.manual_seed(0)
    pls = float(np.exp((np.log(2048) - np.log(16)) / 15))
    net = tcnn.NetworkWithInputEncoding(3, 16, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2,
                                                "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": pls},
                                        {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                                         "n_neurons": 64, "n_hidden_layers": 1}).to(dev)
    with torch.no_grad():
        net.params.uniform_(-1, 1)
    mod = net.native_tcnn_module
    half = net.params.detach().to(torch.float16)
    for label, R, spread in (("train 4096x48 (untrained: samples spread along the ray)", 4096, 1.0),
                             ("train 4096x48 (trained: samples within 2 % of the surface)", 4096, 0.02),
                             ("render 32768x48 (trained)", 32768, 0.02)):
        S = 48
        n = R * S
        o = (torch.rand(R, 1, 3, device=dev) - 0.5) * 1.2
        d = torch.nn.functional.normalize(torch.randn(R, 1, 3, device=dev), dim=-1)
        t0 = 0.3 + 2.5 * torch.rand(R, 1, 1, device=dev)
        t = t0 * (1 + spread * (torch.linspace(-1, 1, S, device=dev).view(1, S, 1)))
        p = o + d * t.clamp_min(0.05)
        mag = p.abs().amax(dim=-1, keepdim=True).clamp_min(1e-9)
        p = torch.where(mag > 1, (2 - 1 / mag) * (p / mag), p)
        x = ((p + 2) / 4).reshape(-1, 3).contiguous()
        out = torch.empty(n, 16, dtype=torch.float16, device=dev)
        ctx = torch.empty(mod.ctx_bytes(n), dtype=torch.uint8, device=dev)
        res = {}
        for form in (0, 4, 0, 4):
            mod.set_option("grid_fwd_small_form", form)
            best = 1e9
            for rnd in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    _lib.check(lib.nvo_fwd(mod.handle, _stream(dev), n, _ptr(x), _ptr(half), _ptr(out), _ptr(ctx)), "nvo_fwd")
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
            enc = ctx[:16 * n * 4].clone()
            if form in res:
                assert torch.equal(res[form], enc)
            res[form] = enc
            print(f"{label:62s} form {form}: grid_fwd + mlp_fwd {best:8.1f} us per pair")
        same = bool(torch.equal(res[0], res[4]))
        print(f"{label:62s} encoded features bit-identical: {same}")
        assert same
    mod.set_option("grid_fwd_small_form", -1)


if __name__ == "__main__":
    main()
