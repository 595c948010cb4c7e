"""Iso-surface extraction behind pyngp's compute_and_save_marching_cubes_mesh (CPU: plain tensor ops): a sphere must
come out closed (every edge shared by exactly two triangles, Euler characteristic 2) at the right radius, and the two
file formats must round-trip."""
import numpy as np
import torch


def _sphere(n=40, radius=0.6):
    ax = torch.linspace(-1, 1, n)
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    return radius - torch.sqrt(X * X + Y * Y + Z * Z)


def test_sphere_is_closed_and_on_the_iso_surface():
    from nerf_vo_amd.meshing import marching_tetrahedra

    v, f = marching_tetrahedra(_sphere(), [-1, -1, -1], [1, 1, 1], 0.0, slab=7)
    r = v.norm(dim=1)
    assert f.shape[0] > 1000 and float(r.min()) > 0.595 and float(r.max()) < 0.6005  # linear cuts of a convex field
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).sort(dim=1).values
    u, c = torch.unique(e, dim=0, return_counts=True)
    assert bool((c == 2).all()), "the surface has holes or non-manifold edges"
    assert v.shape[0] - u.shape[0] + f.shape[0] == 2
    # outward orientation is consistent: every face normal points the same way relative to the centre
    n = torch.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]], dim=1)
    s = (n * v[f].mean(dim=1)).sum(dim=1)
    assert bool((s > 0).all()) or bool((s < 0).all())


def test_empty_field_and_file_formats(tmp_path):
    from nerf_vo_amd.meshing import marching_tetrahedra, write_mesh

    v, f = marching_tetrahedra(torch.zeros(8, 8, 8), [0, 0, 0], [1, 1, 1], 0.5)
    assert v.shape == (0, 3) and f.shape == (0, 3)
    v, f = marching_tetrahedra(_sphere(20), [-1, -1, -1], [1, 1, 1], 0.0)
    write_mesh(str(tmp_path / "m.obj"), v, f)
    write_mesh(str(tmp_path / "m.ply"), v, f)
    lines = open(tmp_path / "m.obj").read().splitlines()
    assert sum(l.startswith("v ") for l in lines) == v.shape[0] and sum(l.startswith("f ") for l in lines) == f.shape[0]
    raw = open(tmp_path / "m.ply", "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert f"element vertex {v.shape[0]}".encode() in head and f"element face {f.shape[0]}".encode() in head
    assert len(body) == v.shape[0] * 12 + f.shape[0] * 13
    back = np.frombuffer(body[: v.shape[0] * 12], dtype="<f4").reshape(-1, 3)
    assert np.allclose(back, v.numpy())
