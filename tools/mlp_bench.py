#!/usr/bin/env python3
"""Fused-MLP kernel timing (HIP-event profiler of the library) vs batch size and workgroup cap.
Usage: python tools/mlp_bench.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

entry.build()
import nerf_vo_amd.tinycudann as tcnn  # noqa: E402
from nerf_vo_amd import _lib  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    shapes = [(64, 16, 64, 2), (32, 16, 64, 1), (10, 1, 16, 1)]
    for n_in, n_out, width, hidden in shapes:
        net = tcnn.Network(n_in, n_out, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                                         "n_neurons": width, "n_hidden_layers": hidden}).to(dev)
        for n in (196608, 98304, 49152):
            x = torch.rand(n, n_in, device=dev, requires_grad=True)
            for cap in (128, 256, 512, 1024):
                os.environ["NVO_MLP_BWD_BLOCKS"] = str(cap)
                os.environ["NVO_MLP_FWD_BLOCKS"] = str(cap)
                for it in range(13):
                    if it == 3:
                        torch.cuda.synchronize()
                        lib.nvo_profile_enable(1)
                    net.params.grad = None
                    x.grad = None
                    net(x).float().sum().backward()
                torch.cuda.synchronize()
                need = lib.nvo_profile_summary(None, 0)
                buf = C.create_string_buffer(int(need) + 16)
                lib.nvo_profile_summary(buf, len(buf))
                lib.nvo_profile_enable(0)
                out = {}
                for line in buf.value.decode().strip().splitlines():
                    name, cnt, total = line.rsplit(",", 2)
                    out[name.split("[")[0]] = float(total) / int(cnt) * 1e3
                print(f"{n_in}->{width}x{hidden}->{n_out}  N={n:7d} cap={cap:5d}  fwd {out.get('mlp_fwd', 0):7.1f} us  bwd {out.get('mlp_bwd', 0):7.1f} us")


if __name__ == "__main__":
    main()
