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


def _pcg_hash(v):
    """PCG output permutation on uint32 arrays (the product's counter-based generator)."""
    v = np.asarray(v, np.uint32)
    state = v * np.uint32(747796405) + np.uint32(2891336453)
    word = ((state >> ((state >> np.uint32(28)) + np.uint32(4))) ^ state) * np.uint32(277803737)
    return (word >> np.uint32(22)) ^ word


def _hash_uniform(seed, step, stream, i):
    with np.errstate(over="ignore"):
        a = _pcg_hash(np.uint32(seed) ^ (np.uint32(stream) * np.uint32(0x9E3779B9)))
        h = _pcg_hash(_pcg_hash(a + np.uint32(step)) + np.asarray(i, np.uint32))
    return (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def morton3d_invert_numpy(idx):
    idx = np.asarray(idx, np.uint32)
    out = []
    for a in range(3):
        v = np.zeros_like(idx)
        for b in range(10):
            v |= ((idx >> np.uint32(3 * b + a)) & np.uint32(1)) << np.uint32(b)
        out.append(v)
    return out


def refresh_samples(n, first, n_total, step, seed, stream_id, n_levels, grid, thresh, aabb_lo, aabb_hi):
    """Samples first .. first + n - 1 of one pass of the density-grid refresh past the warm-up [UPSTREAM instant-ngp
    testbed_nerf.cu generate_grid_samples_nerf_nonuniform; SURVEY.md section 2.4 K16]: a random cascade, the first of ten
    candidate cells ((i + step * n_total) * 56924617 + j * 19349663 + 96925573) mod 128^3 (uint32 wrap-around) whose grid
    value exceeds `thresh` (the tenth when none does), a uniform point inside the cell.  float32 operations in the
    kernel's order.  Returns (cell_idx [n] = level * 128^3 + Morton index, x01 [n,3])."""
    g = np.ascontiguousarray(grid, np.float32).reshape(n_levels, CELLS)
    i = (np.uint32(first) + np.arange(n, dtype=np.uint32)).astype(np.uint32)
    level = (_hash_uniform(seed, step, 4 * stream_id, i) * np.float32(n_levels)).astype(np.uint32)
    level = np.minimum(level, np.uint32(n_levels - 1))
    with np.errstate(over="ignore"):
        base = (i + np.uint32(step) * np.uint32(n_total)) * np.uint32(56924617) + np.uint32(96925573)
        idx = np.zeros(n, np.uint32)
        done = np.zeros(n, bool)
        for j in range(10):
            cand = (base + np.uint32(j) * np.uint32(19349663)) & np.uint32(CELLS - 1)
            idx = np.where(done, idx, cand)
            done |= g[level, idx] > np.float32(thresh)
    c = morton3d_invert_numpy(idx)
    scale = np.ldexp(np.float32(1.0), level.astype(np.int32)).astype(np.float32)
    x01 = np.zeros((n, 3), np.float32)
    lo, hi = np.float32(aabb_lo), np.float32(aabb_hi)
    for a in range(3):
        u = (c[a].astype(np.float32) + _hash_uniform(seed, step, 4 * stream_id + 1 + a, i)) / np.float32(GRID)
        p = (u - np.float32(0.5)) * scale + np.float32(0.5)
        x01[:, a] = np.clip((p - lo) / (hi - lo), np.float32(0.0), np.float32(1.0))
    return level * np.uint32(CELLS) + idx, x01


def cells_seen(cell_idx, n_levels, intrinsics, c2w, H, W, margin=0.0):
    """Which cells (level * 128^3 + Morton index) some camera sees [UPSTREAM instant-ngp mark_untrained_density_grid;
    SURVEY.md section 2.4 K16]: one of the eight corners in front of a camera (cosine to the viewing axis >= 1e-4) and
    projecting strictly inside its image (grown by `margin` projected cell diagonals on every side; 0 = upstream).  Pinhole cameras, OpenGL axes: c2w [F,3,4], intrinsics [F,4] = fx, fy, cx, cy.
    float32 operations in the kernel's order."""
    f32 = np.float32
    cell_idx = np.asarray(cell_idx, np.uint32)
    level = (cell_idx // np.uint32(CELLS)).astype(np.int32)
    c = morton3d_invert_numpy(cell_idx % np.uint32(CELLS))
    scale = np.ldexp(f32(1.0), level).astype(f32)
    size = np.ldexp(f32(1.0 / GRID), level).astype(f32)
    p0 = [((c[a].astype(f32) / f32(GRID) - f32(0.5)) * scale + f32(0.5)).astype(f32) for a in range(3)]
    seen = np.zeros(cell_idx.shape, bool)
    K = np.asarray(intrinsics, f32)
    M = np.asarray(c2w, f32).reshape(-1, 12)
    for j in range(M.shape[0]):
        m = M[j]
        fx, fy, cx, cy = K[j]
        for k in range(8):
            d = [(p0[a] + (size if (k >> a) & 1 else f32(0.0))).astype(f32) - m[4 * a + 3] for a in range(3)]
            qx = (m[0] * d[0] + m[4] * d[1]) + m[8] * d[2]
            qy = (m[1] * d[0] + m[5] * d[1]) + m[9] * d[2]
            qz = (m[2] * d[0] + m[6] * d[1]) + m[10] * d[2]
            depth = -qz
            ln = np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]).astype(f32)
            front = (depth >= f32(1e-4) * ln) & (depth > 0)
            with np.errstate(divide="ignore", invalid="ignore"):
                px = cx + fx * (qx / depth)
                py = cy - fy * (qy / depth)
                mx = f32(margin) * (fx * (size * f32(1.7320508) / depth))
                my = f32(margin) * (fy * (size * f32(1.7320508) / depth))
            seen |= front & (px > -mx) & (py > -my) & (px < f32(W) + mx) & (py < f32(H) + my)
    return seen
