#!/usr/bin/env python3
"""A/B of the proposal grids' parameter backward with / without slice codes (NVO_GRID_SLICE_CODES = 1 | 0; read once per
process): run once per value -- the second run compares its gradients bit for bit with what the first one saved; both
print the launch times.  The module is configured as the engine configures its proposal networks.
Usage: NVO_GRID_SLICE_CODES=0 python tools/probes/bwd_codes_ab.py /tmp/bw.pt ; NVO_GRID_SLICE_CODES=1 python ... /tmp/bw.pt"""
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
from nerf_vo_amd.tinycudann.modules import _ptr, _stream  # noqa: E402


def pls(b, m, L):
    return float(np.exp((np.log(m) - np.log(b)) / (L - 1)))


def main():
    path = sys.argv[1]
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    torch.manual_seed(0)
    outs = {}
    for label, mx, S, dead in (("prop0 all live", 128, 256, 0.0), ("prop1 all live", 256, 96, 0.0), ("prop0 60% dead", 128, 256, 0.6),
                               ("prop0 95% dead", 128, 256, 0.95)):
        R = 4096
        n = R * S
        net = tcnn.NetworkWithInputEncoding(3, 1, {"otype": "HashGrid", "n_levels": 5, "n_features_per_level": 2,
                                                   "log2_hashmap_size": 17, "base_resolution": 16, "per_level_scale": pls(16, mx, 5)},
                                            {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                                             "n_neurons": 16, "n_hidden_layers": 1}).to(dev)
        with torch.no_grad():
            net.params.uniform_(-1, 1)
        mod = net.native_tcnn_module
        for k, v in (("compact_output", 1), ("recompute_hidden", 1), ("grid_bwd_mode", 1), ("grid_acc_bits", 32), ("grid_bwd_runs", 1),
                     ("grid_bwd_batch", n), ("grid_bwd_dense_share", 120), ("grid_compact_live", 1)):
            mod.set_option(k, v)
        o = (torch.rand(R, 1, 3, device=dev) - 0.5) * 1.2
        d = torch.nn.functional.normalize(torch.randn(R, 1, 3, device=dev), dim=-1)
        t = 1.0 / torch.linspace(1.0 / 0.05, 1.0 / 30.0, S, device=dev).view(1, S, 1)
        p = o + d * t
        mag = p.abs().amax(dim=-1, keepdim=True).clamp_min(1e-9)
        p = torch.where(mag > 1, (2 - 1 / mag) * (p / mag), p)
        x = ((p + 2) / 4).reshape(-1, 3).contiguous()
        half = net.params.detach().to(torch.float16)
        out = torch.empty(n, dtype=torch.float16, device=dev)
        dout = (torch.randn(n, device=dev) * 64).to(torch.float16)
        if dead > 0:  # dead samples come in runs along the rays (what the proposal losses produce)
            runs = (torch.rand((n + 15) // 16, device=dev) < dead).repeat_interleave(16)[:n]
            dout[runs] = 0
        ctx = torch.empty(mod.ctx_bytes(n), dtype=torch.uint8, device=dev)
        grads = torch.zeros(mod.n_params, dtype=torch.float32, device=dev)
        _lib.check(lib.nvo_fwd(mod.handle, _stream(dev), n, _ptr(x), _ptr(half), _ptr(out), _ptr(ctx)), "nvo_fwd")
        for it in range(13):
            if it == 3:
                torch.cuda.synchronize()
                lib.nvo_profile_enable(1)
            _lib.check(lib.nvo_bwd(mod.handle, _stream(dev), n, _ptr(x), _ptr(half), _ptr(out), _ptr(dout), _ptr(ctx), None,
                                   _ptr(grads)), "nvo_bwd")
        torch.cuda.synchronize()
        need = lib.nvo_profile_summary(None, 0)
        buf = C.create_string_buffer(int(need) + 16)
        lib.nvo_profile_summary(buf, len(buf))
        lib.nvo_profile_enable(0)
        for line in buf.value.decode().strip().splitlines():
            name, cnt, total = line.rsplit(",", 2)
            if name.startswith("grid_bwd"):
                print(f"{label:16s} N={n:8d} {name:20s} avg {float(total) / int(cnt) * 1e3:8.1f} us  (NVO_GRID_SLICE_CODES={os.environ.get('NVO_GRID_SLICE_CODES', 'default')})")
        outs[label] = grads.cpu()
    if os.path.exists(path):
        ref = torch.load(path)
        for k, v in outs.items():
            grid = v[512:]  # (the MLP's 512 weights lead the block: float atomics between workgroups, not bitwise stable)
            same = bool(torch.equal(ref[k][512:].view(torch.int32), grid.view(torch.int32)))
            rel = float((ref[k][512:] - grid).abs().max() / ref[k][512:].abs().max())
            print(f"{k:16s} grid gradient bit-identical to the saved run: {same} (largest difference / largest gradient {rel:.2e})")
    else:
        torch.save(outs, path)
        print(f"saved {path}")


if __name__ == "__main__":
    main()
