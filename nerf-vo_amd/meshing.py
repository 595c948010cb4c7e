"""Iso-surface extraction for the occupancy-grid back-end's ``compute_and_save_marching_cubes_mesh`` (the call the
reference makes at /root/reference/evaluation/nerf_renderer.py:296-300; mesh evaluation itself is SURVEY.md section 2
out of scope).  Off the hot path: plain torch tensor ops on whatever device holds the density samples.

The surface is extracted with MARCHING TETRAHEDRA -- every grid cube is cut into the six tetrahedra around its main
diagonal and each tetrahedron contributes zero, one or two triangles from its four corner signs -- not with the
256-case marching-cubes tables [UPSTREAM instant-ngp marching_cubes.cu, not vendored]: same iso-surface up to the
triangulation, no ambiguous cases, and the case table below is generated, not copied.
"""
from __future__ import annotations

import struct

import numpy as np
import torch

# cube corners (x, y, z offsets) and the six tetrahedra that share the diagonal 0-6
_CORNERS = torch.tensor([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]])
_TETS = torch.tensor([[0, 5, 1, 6], [0, 1, 2, 6], [0, 2, 3, 6], [0, 3, 7, 6], [0, 7, 4, 6], [0, 4, 5, 6]])


def _case_table():
    """[16][2 triangles][3 points][2 tetrahedron corners of the cut edge]; -1 = no triangle."""
    table = -torch.ones(16, 2, 3, 2, dtype=torch.long)
    for case in range(16):
        inside = [k for k in range(4) if (case >> k) & 1]
        outside = [k for k in range(4) if not (case >> k) & 1]
        if len(inside) == 1 or len(inside) == 3:
            a = inside[0] if len(inside) == 1 else outside[0]
            others = [k for k in range(4) if k != a]
            tri = [[a, others[0]], [a, others[1]], [a, others[2]]]
            if len(inside) == 3:
                tri = [tri[0], tri[2], tri[1]]
            table[case, 0] = torch.tensor(tri)
        elif len(inside) == 2:
            a, b = inside
            c, d = outside
            table[case, 0] = torch.tensor([[a, c], [a, d], [b, d]])
            table[case, 1] = torch.tensor([[a, c], [b, d], [b, c]])
    return table


_TABLE = _case_table()
# one corner on the inside of the surface per case (orientation reference; -1: no surface)
_INSIDE_REF = torch.tensor([next((k for k in range(4) if (case >> k) & 1), -1) if 0 < case < 15 else -1 for case in range(16)])


@torch.no_grad()
def marching_tetrahedra(values: torch.Tensor, lower, upper, threshold: float, slab: int = 16):
    """values [nx, ny, nz] sampled at the corners of a regular grid spanning [lower, upper] -> (vertices [V, 3] float32,
    faces [F, 3] int64).  Vertices shared by neighbouring triangles are merged."""
    dev = values.device
    nx, ny, nz = values.shape
    lo = torch.as_tensor(lower, dtype=torch.float32, device=dev)
    hi = torch.as_tensor(upper, dtype=torch.float32, device=dev)
    step = (hi - lo) / torch.tensor([max(nx - 1, 1), max(ny - 1, 1), max(nz - 1, 1)], dtype=torch.float32, device=dev)
    corners, tets, table = _CORNERS.to(dev), _TETS.to(dev), _TABLE.to(dev)
    tris = []
    for z0 in range(0, nz - 1, slab):
        z1 = min(nz - 1, z0 + slab)
        v = values[:, :, z0:z1 + 1].float()
        inside = v > threshold
        # cubes of the slab whose corners are not all on one side
        c = torch.stack([inside[dx:nx - 1 + dx, dy:ny - 1 + dy, dz:z1 - z0 + dz] for dx, dy, dz in corners.tolist()], dim=-1)
        mixed = c.any(dim=-1) & ~c.all(dim=-1)
        cube = mixed.nonzero()  # [M, 3] (x, y, z - z0)
        if cube.numel() == 0:
            continue
        pos = cube[:, None, :] + corners[None, :, :]  # [M, 8, 3] grid coordinates of the corners
        val = v[pos[..., 0], pos[..., 1], pos[..., 2]]  # [M, 8]
        pos = pos.clone()
        pos[..., 2] += z0
        tv = val[:, tets]  # [M, 6, 4]
        tp = pos[:, tets]  # [M, 6, 4, 3]
        case = ((tv > threshold).long() * torch.tensor([1, 2, 4, 8], device=dev)).sum(-1)  # [M, 6]
        edges = table[case]  # [M, 6, 2, 3, 2]
        valid = edges[..., 0, 0] >= 0  # [M, 6, 2]
        e = edges.clamp(min=0)
        M = tv.shape[0]
        idx_m = torch.arange(M, device=dev)[:, None, None, None].expand(M, 6, 2, 3)
        idx_t = torch.arange(6, device=dev)[None, :, None, None].expand(M, 6, 2, 3)
        va, vb = tv[idx_m, idx_t, e[..., 0]], tv[idx_m, idx_t, e[..., 1]]
        pa, pb = tp[idx_m, idx_t, e[..., 0]].float(), tp[idx_m, idx_t, e[..., 1]].float()
        t = ((threshold - va) / (vb - va)).clamp(0.0, 1.0).unsqueeze(-1)
        pts = pa + t * (pb - pa)  # grid coordinates [M, 6, 2, 3, 3]
        # consistent winding: the normal points away from a corner that lies inside the surface
        ref = _INSIDE_REF.to(dev)[case].clamp(min=0)  # [M, 6]
        pin = tp[torch.arange(M, device=dev)[:, None], torch.arange(6, device=dev)[None, :], ref].float()  # [M, 6, 3]
        nrm = torch.cross(pts[..., 1, :] - pts[..., 0, :], pts[..., 2, :] - pts[..., 0, :], dim=-1)  # [M, 6, 2, 3]
        out = pts.mean(dim=-2) - pin[:, :, None, :]
        flip = (nrm * out).sum(-1) < 0  # [M, 6, 2]
        p1 = torch.where(flip[..., None], pts[..., 2, :], pts[..., 1, :])
        p2 = torch.where(flip[..., None], pts[..., 1, :], pts[..., 2, :])
        pts = torch.stack([pts[..., 0, :], p1, p2], dim=-2)
        tris.append(pts[valid])
    if not tris:
        return torch.zeros(0, 3, device=dev), torch.zeros(0, 3, dtype=torch.long, device=dev)
    soup = torch.cat(tris).reshape(-1, 3)  # [3 F, 3] grid coordinates
    # merge shared vertices (cut points lie on grid edges: quantise to 1/4096 of a cell)
    key = torch.round(soup * 4096.0).long()
    uniq, inverse = torch.unique(key, dim=0, return_inverse=True)
    verts = torch.zeros(uniq.shape[0], 3, device=dev).index_copy_(0, inverse, soup)
    faces = inverse.view(-1, 3)
    faces = faces[(faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])]
    return lo + verts * step, faces


def write_mesh(path: str, vertices: torch.Tensor, faces: torch.Tensor) -> None:
    """.obj (text) or .ply (binary little endian), by extension."""
    v = vertices.detach().cpu().numpy().astype(np.float32)
    f = faces.detach().cpu().numpy().astype(np.int32)
    if path.lower().endswith(".obj"):
        with open(path, "w") as fh:
            fh.write("# nerf_vo_amd marching-tetrahedra mesh\n")
            for p in v:
                fh.write(f"v {p[0]:.6f} {p[1]:.6f} {p[2]:.6f}\n")
            for t in f:
                fh.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")
        return
    with open(path, "wb") as fh:
        fh.write(("ply\nformat binary_little_endian 1.0\ncomment nerf_vo_amd marching-tetrahedra mesh\n"
                  f"element vertex {v.shape[0]}\nproperty float x\nproperty float y\nproperty float z\n"
                  f"element face {f.shape[0]}\nproperty list uchar int vertex_indices\nend_header\n").encode())
        fh.write(v.tobytes())
        rec = np.empty(f.shape[0], dtype=np.dtype([("n", "u1"), ("i", "<i4", 3)]))
        rec["n"] = 3
        rec["i"] = f
        fh.write(rec.tobytes())
