"""TEST INFRASTRUCTURE -- CPU restatement of the reference's keyframe depth alignment (SURVEY.md section 8f
row f2): monocular depth maps are brought to the metric scale of DPVO's sparse patches by a per-frame
scale/shift, after a quantile-based outlier removal of the patches.

Follows /root/reference/nerf_vo/enhancement/enhancement_module.py:
  * dpvo_remove_outliers  <- :131-146  (1/12 and 11/12 quantiles of the centre inverse depth, tie-breaking noise,
                                        global boolean-mask compaction + reshape, and the `except` fallback)
  * align_depth           <- :61-99    (centre pixel of every patch, x4 pixel coordinates, depth = clip(1/inv, 0, 5),
                                        scale = std ratio, shift = frame mean x (mean ratio - scale), clip to [0, 5])

PARITY PINNED: unlike the rest of oracle/, this file is checked against outputs of the reference itself
(tests/golden/enhancement_golden.npz, produced by tests/golden/make_golden_enhancement.py, which imports
/root/reference/nerf_vo/enhancement/enhancement_module.py and runs its step() on CPU).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from __future__ import annotations

import torch


def dpvo_remove_outliers(dpvo_patches: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """dpvo_patches [K,M,3,P,P] (x/4, y/4, inverse depth), noise [K,M,1,1] in [0,1) (the reference draws it with
    torch.rand; injected here so that runs are reproducible).  Returns [K, int(M*5/6), 3, P, P], or -- when the
    number of survivors does not allow that reshape -- the reference's fallback: ALL patches, with every element
    < 1e-3 replaced by the global mean."""
    p = dpvo_patches.clone()
    p[:, :, 2] += noise * 1e-4
    centre = p[:, :, 2, 1, 1]
    lo = torch.quantile(centre, q=1 / 12, dim=1)[:, None]
    hi = torch.quantile(centre, q=11 / 12, dim=1)[:, None]
    outlier = (centre < lo) | (centre > hi)
    keep = int(p.shape[1] * 5 / 6)
    kept = p[~outlier]
    if kept.shape[0] == p.shape[0] * keep:
        return kept.reshape(p.shape[0], keep, 3, p.shape[3], p.shape[4])
    p[p < 1e-3] = torch.mean(p)
    return p


def align_depth(frames_depth: torch.Tensor, patches: torch.Tensor) -> torch.Tensor:
    """frames_depth [K,1,H,W] monocular depth; patches [K,M',3,P,P] after outlier removal.  Returns the aligned,
    clipped depth [K,1,H,W]."""
    c = patches[:, :, :, 1, 1].clone()
    c[:, :, :2] = c[:, :, :2] * 4
    c[:, :, 2] = 1 / c[:, :, 2]
    c[:, :, 2] = c[:, :, 2].clip(0, 5)
    K, M = c.shape[0], c.shape[1]
    rows = torch.arange(0, K, dtype=torch.long).repeat(M, 1).t().reshape(-1)
    at = frames_depth[rows, 0, c[:, :, 1].reshape(-1).long(), c[:, :, 0].reshape(-1).long()].reshape(K, M)
    sparse = c[:, :, 2]
    scale = (torch.std(sparse, dim=1) / torch.std(at, dim=1))[:, None, None, None]
    shift = torch.mean(frames_depth, dim=[1, 2, 3], keepdim=True) * (
        (torch.mean(sparse, dim=1) / torch.mean(at, dim=1))[:, None, None, None] - scale)
    return torch.clip(frames_depth * scale + shift, 0, 5)


def enhance_depth(frames_depth: torch.Tensor, dpvo_patches: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    return align_depth(frames_depth, dpvo_remove_outliers(dpvo_patches, noise))
