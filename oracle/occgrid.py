"""CPU oracle: cascaded Morton-bitfield occupancy grid + DDA ray marcher (TEST INFRASTRUCTURE; parity
unpinned).  Thin numpy wrapper over oracle/c/nvo_oracle.c, which restates instant-ngp's NeRF testbed
marcher (the back-end behind `mapping_module: 'instant-ngp'`,
/root/reference/nerf_vo/mapping/instant_ngp.py:33-50,104-105).  All
occupancy decisions are integer/branch results of IEEE float ops evaluated in the same order as the
HIP kernel (no fma contraction), so tests compare them bit for bit."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .grid import _clib

GRID = 128
CELLS = GRID ** 3


def _lib():
    lib = _clib()
    if not getattr(lib, "_occ_ready", False):
        lib.nvo_oracle_morton3d.restype = C.c_uint32
        lib.nvo_oracle_morton3d.argtypes = [C.c_uint32] * 3
        lib.nvo_oracle_occ_march_ray.restype = C.c_uint32
        lib.nvo_oracle_occ_march_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                                 C.c_float, C.c_void_p, C.c_void_p, C.c_uint32]
        lib.nvo_oracle_occ_bitfield.restype = None
        lib.nvo_oracle_occ_bitfield.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
        lib._occ_ready = True
    return lib


def morton3d(x: int, y: int, z: int) -> int:
    return int(_lib().nvo_oracle_morton3d(x, y, z))


def morton3d_numpy(x, y, z):
    """Independent restatement: bit interleave by explicit loops."""
    x, y, z = (np.asarray(v, dtype=np.uint32) for v in (x, y, z))
    out = np.zeros_like(x)
    for b in range(10):
        out |= ((x >> b) & 1) << (3 * b)
        out |= ((y >> b) & 1) << (3 * b + 1)
        out |= ((z >> b) & 1) << (3 * b + 2)
    return out


def march_rays(origins, directions, bitfield, n_levels, cone_angle, t_near, jitter, max_out=1024):
    """origins/directions [R,3] float32 (normalised frame, unit directions), bitfield uint8
    [n_levels, CELLS/8], jitter [R].  Returns (counts [R], t [R,max_out], dt [R,max_out])."""
    o = np.ascontiguousarray(origins, np.float32)
    d = np.ascontiguousarray(directions, np.float32)
    bf = np.ascontiguousarray(bitfield, np.uint8)
    R = o.shape[0]
    counts = np.zeros(R, np.uint32)
    t = np.zeros((R, max_out), np.float32)
    dt = np.zeros((R, max_out), np.float32)
    lib = _lib()
    for r in range(R):
        counts[r] = lib.nvo_oracle_occ_march_ray(o[r].ctypes.data, d[r].ctypes.data, bf.ctypes.data, n_levels,
                                                 C.c_float(cone_angle), C.c_float(t_near), C.c_float(float(jitter[r])),
                                                 t[r].ctypes.data, dt[r].ctypes.data, max_out)
    return counts, t, dt


def grid_to_bitfield(grid, n_levels, threshold=0.01):
    g = np.ascontiguousarray(grid, np.float32).reshape(n_levels, CELLS)
    out = np.zeros((n_levels, CELLS // 8), np.uint8)
    _lib().nvo_oracle_occ_bitfield(g.ctypes.data, n_levels, C.c_float(threshold), out.ctypes.data)
    return out


def ema_update(grid, new_values, decay=0.95):
    """instant-ngp ema_grid_samples_nerf: negative (never-visible) cells stay, others max(decay*old, new)."""
    g = np.asarray(grid, np.float32)
    return np.where(g < 0, g, np.maximum(g * np.float32(decay), np.asarray(new_values, np.float32)))
