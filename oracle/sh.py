"""CPU oracle: real spherical harmonics direction encoding, degree <= 4 (TEST INFRASTRUCTURE;
parity unpinned).  Restates tiny-cuda-nn's SphericalHarmonics encoding (upstream
encodings/spherical_harmonics.h; input is (d+1)/2, mapped back with
2*x-1), used by NerfactoField's direction encoding (reference call site
/root/reference/nerf_vo/mapping/nerfstudio.py:151)."""
from __future__ import annotations

import torch


def sh_encode(d01: torch.Tensor, degree: int = 4) -> torch.Tensor:
    x, y, z = (d01[:, i] * 2.0 - 1.0 for i in range(3))
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    o = [torch.full_like(x, 0.28209479177387814)]
    if degree > 1:
        o += [-0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x]
    if degree > 2:
        o += [1.0925484305920792 * xy, -1.0925484305920792 * yz,
              0.94617469575755997 * z2 - 0.31539156525251999, -1.0925484305920792 * xz,
              0.54627421529603959 * x2 - 0.54627421529603959 * y2]
    if degree > 3:
        o += [0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z,
              0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0),
              0.45704579946446572 * x * (1.0 - 5.0 * z2), 1.4453057213202769 * z * (x2 - y2),
              0.59004358992664352 * x * (-x2 + 3.0 * y2)]
    return torch.stack(o, dim=1)
