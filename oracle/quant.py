"""CPU oracle (TEST INFRASTRUCTURE; parity unpinned): emulation of the 16-bit storage points of the HIP kernels.

The kernels keep activations, weights and gradients in a 16-bit format between (and inside) the fused MLPs and
accumulate in fp32.  Two formats exist: fp16 (tiny-cuda-nn's precision; the reference trains with
mixed_precision=True, /root/reference/nerf_vo/mapping/nerfstudio.py:59) and bfloat16 (BASELINE.json configs[4]:
"MFMA bf16 MLP + fp32 hash accumulate").  ``q16`` rounds to the ACTIVE format (round-to-nearest-even, as
v_cvt_f16_f32 / v_cvt_pk_bf16_f32 do) with a straight-through gradient; ``activation_format`` switches it.
"""
from __future__ import annotations

import contextlib

import torch

_FORMAT = "f16"
_DTYPES = {"f16": torch.float16, "bf16": torch.bfloat16}


def active_format() -> str:
    return _FORMAT


@contextlib.contextmanager
def activation_format(fmt: str):
    """with activation_format("bf16"): ... -> every q16() inside rounds to bfloat16."""
    global _FORMAT
    if fmt not in _DTYPES:
        raise ValueError(fmt)
    old, _FORMAT = _FORMAT, fmt
    try:
        yield
    finally:
        _FORMAT = old


def q16(x: torch.Tensor, fmt: str | None = None) -> torch.Tensor:
    """Round to the 16-bit format (value), identity (gradient)."""
    dt = _DTYPES[fmt or _FORMAT]
    # float64 -> 16 bit directly would round once, as the hardware conversion from the fp32 accumulator does up to
    # the fp32 rounding of the accumulator itself (double rounding differences are below every tolerance used)
    return x + (x.to(torch.float32).to(dt).to(x.dtype) - x).detach()
