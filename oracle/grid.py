"""CPU oracle: multiresolution hash-grid encoding (TEST INFRASTRUCTURE; parity unpinned).

Restates tiny-cuda-nn's HashGrid encoding (upstream encodings/grid.h: kernel_grid,
kernel_grid_backward, kernel_grid_backward_input -- SURVEY.md section 2.4 K1-K3), reached by the
reference through nerfstudio's fields (/root/reference/nerf_vo/mapping/nerfstudio.py:151,
/root/reference/nerf_vo/mapping/nerfstudio_utils.py:333-350).  Integer paths come from
oracle/c/nvo_oracle.c (same libm as the product's host code) with an independent numpy restatement
for cross-checking; the float path is torch-CPU float64 with autograd providing both gradients.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from dataclasses import dataclass
from pathlib import Path

import numpy as np
import torch

_HERE = Path(__file__).resolve().parent
_LIB = None

PRIMES = (np.uint32(1), np.uint32(2654435761), np.uint32(805459861))


def _clib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        so = _HERE / "libnvo_oracle.so"
        src = _HERE / "c" / "nvo_oracle.c"
        if not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
            subprocess.run(["make", "-C", str(_HERE), "-s"], check=True)
        _LIB = C.CDLL(str(so))
        _LIB.nvo_oracle_level_table.restype = C.c_uint32
        _LIB.nvo_oracle_level_table.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p]
        _LIB.nvo_oracle_grid_indices.restype = None
        _LIB.nvo_oracle_grid_indices.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]
    return _LIB


@dataclass
class GridSpec:
    n_levels: int
    n_features: int
    log2_hashmap_size: int
    base_resolution: int
    per_level_scale: float
    levels: np.ndarray  # uint32 [L,4] = offset, size, resolution, hashed
    scales: np.ndarray  # float32 [L]

    @property
    def n_entries(self) -> int:
        return int(self.levels[-1, 0] + self.levels[-1, 1])

    @property
    def n_params(self) -> int:
        return self.n_entries * self.n_features

    @property
    def n_output_dims(self) -> int:
        return self.n_levels * self.n_features


def make_grid_spec(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16,
                   per_level_scale=2.0) -> GridSpec:
    """Level table exactly as tcnn's GridEncodingTemplated constructor builds it (C, glibc libm)."""
    levels = np.zeros((n_levels, 4), dtype=np.uint32)
    scales = np.zeros(n_levels, dtype=np.float32)
    _clib().nvo_oracle_level_table(n_levels, log2_hashmap_size, base_resolution, C.c_float(per_level_scale),
                                   levels.ctypes.data_as(C.c_void_p), scales.ctypes.data_as(C.c_void_p))
    return GridSpec(n_levels, n_features, log2_hashmap_size, base_resolution, float(per_level_scale), levels, scales)


def level_table_numpy(n_levels, log2_hashmap_size, base_resolution, per_level_scale):
    """Independent restatement of the level rule in numpy float32 (cross-check of the C version)."""
    f32 = np.float32
    log2_pls = np.log2(f32(per_level_scale), dtype=f32)
    levels = np.zeros((n_levels, 4), dtype=np.uint32)
    scales = np.zeros(n_levels, dtype=f32)
    offset = 0
    for l in range(n_levels):
        scale = f32(np.exp2(f32(l) * log2_pls, dtype=f32) * f32(base_resolution) - f32(1.0))
        res = int(np.ceil(scale)) + 1
        n = res ** 3
        n = min((n + 7) // 8 * 8, 1 << log2_hashmap_size)
        levels[l] = (offset, n, res, 1 if res ** 3 > n else 0)
        scales[l] = scale
        offset += n
    return levels, scales


def grid_indices_c(spec: GridSpec, x: np.ndarray):
    """[L,N,8] uint32 level-relative corner indices and fp32 trilinear weights (C oracle)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.shape[0]
    idx = np.zeros((spec.n_levels, n, 8), dtype=np.uint32)
    w = np.zeros((spec.n_levels, n, 8), dtype=np.float32)
    _clib().nvo_oracle_grid_indices(spec.n_levels, spec.levels.ctypes.data_as(C.c_void_p),
                                    spec.scales.ctypes.data_as(C.c_void_p), n, x.ctypes.data_as(C.c_void_p),
                                    idx.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p))
    return idx, w


def grid_indices_numpy(spec: GridSpec, x: np.ndarray) -> np.ndarray:
    """Independent numpy uint32 restatement of grid_index (dense stride index / prime-xor hash)."""
    x = np.asarray(x, dtype=np.float32)
    n = x.shape[0]
    out = np.zeros((spec.n_levels, n, 8), dtype=np.uint32)
    with np.errstate(over="ignore"):
        for l in range(spec.n_levels):
            _, size, res, hashed = (int(v) for v in spec.levels[l])
            scale = spec.scales[l]
            # fma(scale, x, 0.5) == round-once of the exact product-sum: evaluate in float64 and round
            pos = (scale.astype(np.float64) * x.astype(np.float64) + 0.5).astype(np.float32)
            cell = np.floor(pos).astype(np.int32).astype(np.uint32)
            for k in range(8):
                p = [cell[:, d] + np.uint32((k >> d) & 1) for d in range(3)]
                if hashed:
                    idx = (p[0] * PRIMES[0]) ^ (p[1] * PRIMES[1]) ^ (p[2] * PRIMES[2])
                else:
                    # (uint32 stride arithmetic like tcnn's grid_index: res * res wraps for res >= 65536)
                    idx = p[0] + p[1] * np.uint32(res & 0xFFFFFFFF) + p[2] * np.uint32((res * res) & 0xFFFFFFFF)
                out[l, :, k] = idx % np.uint32(size)
    return out


def grid_encode(spec: GridSpec, x: torch.Tensor, table: torch.Tensor, quantize_output: bool = False) -> torch.Tensor:
    """Differentiable encode: x [N,3] (any float dtype, may require grad), table [n_entries, F] ->
    [N, L*F] in table.dtype.  Corner indices are taken from the fp32 C oracle (bit-exact integer
    path); the fractional weights are recomputed differentiably in the table's dtype from the same
    fp32 cell decision, so autograd yields d/dtable (scatter) and d/dx (K3)."""
    n = x.shape[0]
    x32 = x.detach().to(torch.float32).cpu().numpy()
    idx, _ = grid_indices_c(spec, x32)
    dt = table.dtype
    outs = []
    for l in range(spec.n_levels):
        off = int(spec.levels[l, 0])
        scale32 = spec.scales[l]
        pos32 = np.empty_like(x32)
        for d in range(3):
            pos32[:, d] = (scale32.astype(np.float64) * x32[:, d].astype(np.float64) + 0.5).astype(np.float32)
        cell = torch.from_numpy(np.floor(pos32).astype(np.float64)).to(dt)
        pos = x.to(dt) * float(scale32) + 0.5
        frac = pos - cell  # differentiable w.r.t. x
        acc = torch.zeros((n, spec.n_features), dtype=dt)
        ids = torch.from_numpy(idx[l].astype(np.int64)) + off
        for k in range(8):
            w = torch.ones(n, dtype=dt)
            for d in range(3):
                w = w * (frac[:, d] if (k >> d) & 1 else (1.0 - frac[:, d]))
            acc = acc + w[:, None] * table[ids[:, k]]
        outs.append(acc)
    y = torch.cat(outs, dim=1)
    if quantize_output:
        from .quant import q16

        y = q16(y)  # straight-through rounding to the network's input format (fp16 | bf16, quant.py)
    return y
