"""CPU oracle: tiny-cuda-nn FullyFusedMLP (TEST INFRASTRUCTURE; parity unpinned).

Restates upstream networks/fully_fused_mlp.cu semantics: bias-free
dense layers, weights row-major [out][in] in layer order, input padded to a multiple of 16 (with
1.0 for raw inputs -- tcnn's identity encoding -- and 0.0 behind a grid encoding), hidden
activation on every hidden layer, separate output activation, output padded to a multiple of 16.
Reached by the reference via nerfstudio's MLP/field classes
(/root/reference/nerf_vo/mapping/nerfstudio.py:151).  torch-CPU float64 + autograd.
"""
from __future__ import annotations

import torch


def pad16(n: int) -> int:
    return (n + 15) // 16 * 16


def mlp_n_params(n_in: int, n_out: int, width: int, n_hidden: int) -> int:
    return width * pad16(n_in) + (n_hidden - 1) * width * width + pad16(n_out) * width


def split_weights(params: torch.Tensor, n_in: int, n_out: int, width: int, n_hidden: int):
    """Flat parameter vector -> list of [out, in] matrices (tcnn layer order)."""
    in_pad, out_pad = pad16(n_in), pad16(n_out)
    shapes = [(width, in_pad)] + [(width, width)] * (n_hidden - 1) + [(out_pad, width)]
    ws, o = [], 0
    for r, c in shapes:
        ws.append(params[o:o + r * c].view(r, c))
        o += r * c
    assert o == params.numel()
    return ws


def _act(name: str, x: torch.Tensor) -> torch.Tensor:
    if name == "ReLU":
        return torch.relu(x)
    if name == "Sigmoid":
        return torch.sigmoid(x)
    if name == "None":
        return x
    raise ValueError(name)


from .quant import q16 as _q16  # 16-bit storage emulation in the ACTIVE format (fp16 | bf16), see quant.py


def mlp_forward(x: torch.Tensor, weights, activation="ReLU", output_activation="None", pad_value=1.0,
                emulate_fp16=True):
    """x [B, n_in] -> [B, out_pad].  With emulate_fp16 the input, hidden activations and output are
    rounded to fp16 where the HIP kernel stores fp16 (accumulation stays exact)."""
    in_pad = weights[0].shape[1]
    b, n_in = x.shape
    if n_in < in_pad:
        x = torch.cat([x, torch.full((b, in_pad - n_in), pad_value, dtype=x.dtype)], dim=1)
    h = _q16(x) if emulate_fp16 else x
    for i, w in enumerate(weights):
        z = h @ w.t()
        last = i == len(weights) - 1
        h = _act(output_activation if last else activation, z)
        if emulate_fp16:
            h = _q16(h)
    return h
