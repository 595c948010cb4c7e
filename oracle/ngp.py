"""CPU oracle: one training step of the occupancy-grid ("instant-ngp") back-end (TEST INFRASTRUCTURE;
parity unpinned).  Restates Testbed::train_nerf_step of NVlabs instant-ngp as forked by NeRF-SLAM
(reference call sites /root/reference/nerf_vo/mapping/instant_ngp.py:33-50,104-105; SURVEY.md section
3.4): occupancy marching (oracle/c/nvo_oracle.c), hash grid + density MLP, SH + rgb MLP, exponential
density / logistic rgb, front-to-back compositing, L2 rgb + L2 depth loss.  torch-CPU float64 with
autograd; fp16 rounding emulated where the kernels store fp16."""
from __future__ import annotations

import numpy as np
import torch

from . import grid as G
from . import mlp as M
from . import occgrid as O
from . import sh as S


from .quant import q16 as _q16  # 16-bit storage emulation in the ACTIVE format (fp16 | bf16), see quant.py


class NgpOracle:
    def __init__(self, aabb_scale=4, desired_resolution=2048, cone_angle=1 / 256, near=0.1, rgb_mult=1.0,
                 depth_mult=1.0, emulate_fp16=True):
        self.aabb_scale = aabb_scale
        self.n_levels = int(np.ceil(np.log2(aabb_scale))) + 1 if aabb_scale > 1 else 1
        pls = float(np.exp((np.log(desired_resolution * aabb_scale) - np.log(16)) / 15))
        self.spec = G.make_grid_spec(16, 2, 19, 16, pls)
        self.cone_angle, self.near = cone_angle, near
        self.rgb_mult, self.depth_mult = rgb_mult, depth_mult
        self.emulate_fp16 = emulate_fp16
        self.lo, self.hi = 0.5 - 0.5 * aabb_scale, 0.5 + 0.5 * aabb_scale
        self.params = {}

    def march(self, origins, directions, bitfield, jitter):
        return O.march_rays(origins.float().numpy(), directions.float().numpy(), bitfield, self.n_levels,
                            self.cone_angle, self.near, np.zeros(len(origins)) if jitter is None else jitter.numpy())

    def forward(self, origins, directions, counts, t, dt, background=None, sh_directions=None, kept=None):
        """Packed forward for the samples found by march().  Returns per-ray rgb, depth, accumulation.
        ``sh_directions``: directions used for the SH encoding (default: ``directions``); the extrinsics
        optimiser differentiates the sample positions only, its test passes a detached copy here.
        ``kept`` [R] (training on the compacted batch, alive_counts() below): ray r is composited over its first kept[r]
        samples only, and sees the background only when nothing was cut (kept[r] == counts[r])
        [UPSTREAM compute_loss_kernel_train_nerf: `if (T < EPSILON) break` / `if (compacted_numsteps == numsteps)
        rgb_ray += T * background_color`]."""
        full_counts = counts
        if kept is not None:
            counts = np.asarray(kept).astype(counts.dtype)
        P = self.params
        R = origins.shape[0]
        ray_idx = np.repeat(np.arange(R), counts.astype(np.int64))
        tt = torch.from_numpy(np.concatenate([t[r, :counts[r]] for r in range(R)]).astype(np.float32)).double()
        dd = torch.from_numpy(np.concatenate([dt[r, :counts[r]] for r in range(R)]).astype(np.float32)).double()
        ri = torch.from_numpy(ray_idx)
        # the kernel evaluates o + t d in fp32
        pos = (origins.float()[ri] + directions.float()[ri] * tt.float()[:, None]).double()
        x01 = ((pos - self.lo) * (1.0 / (self.hi - self.lo))).clamp(0.0, 1.0)
        enc = G.grid_encode(self.spec, x01, P["grid"], quantize_output=self.emulate_fp16)
        dws = M.split_weights(P["density_mlp"], 32, 16, 64, 1)
        dens_out = M.mlp_forward(enc, dws, "ReLU", "None", pad_value=0.0, emulate_fp16=self.emulate_fp16)
        sh = S.sh_encode(((directions if sh_directions is None else sh_directions) + 1.0) / 2.0, 4)
        if self.emulate_fp16:
            sh = _q16(sh)
        rin = torch.cat([dens_out, sh[ri]], dim=-1)
        rws = M.split_weights(P["rgb_mlp"], 32, 3, 64, 2)
        rgb_pre = M.mlp_forward(rin, rws, "ReLU", "None", pad_value=1.0, emulate_fp16=self.emulate_fp16)[:, :3]
        sigma = torch.exp(dens_out[:, 0])
        rgb = torch.sigmoid(rgb_pre)
        out_rgb, out_depth, out_acc = [], [], []
        off = 0
        for r in range(R):
            n = int(counts[r])
            s, c, ts, ds = sigma[off:off + n], rgb[off:off + n], tt[off:off + n], dd[off:off + n]
            off += n
            ddens = s * ds
            T = torch.exp(-(torch.cumsum(ddens, 0) - ddens))
            w = (1 - torch.exp(-ddens)) * T
            Tf = torch.exp(-ddens.sum()) if n else torch.ones((), dtype=torch.float64)
            bg = background[r] if background is not None else torch.zeros(3, dtype=torch.float64)
            if int(full_counts[r]) != n:  # (cut at the transmittance threshold)
                bg = torch.zeros(3, dtype=torch.float64)
            out_rgb.append((w[:, None] * c).sum(0) + Tf * bg)
            out_depth.append((w * ts).sum())
            out_acc.append(w.sum())
        return torch.stack(out_rgb), torch.stack(out_depth), torch.stack(out_acc)

    def loss_dict(self, rgb, depth, gt_rgb, gt_depth, directions_norm, gt_depth_cov=None):
        """L2 rgb + depth term.  ``gt_depth_cov`` (per ray; what update_training_images received as ``depths_cov`` times
        ``depth_cov_scale``, /root/reference/nerf_vo/mapping/instant_ngp.py:77-100): the VARIANCE of the depth target
        (DROID-SLAM's marginal depth covariance, /root/reference/nerf_vo/tracking/droid_slam.py:716-725: z_cov / idepth^4).
        Form chosen [UPSTREAM NeRF-SLAM fork of instant-ngp, not vendored -- spec by the paper's mapping loss
        L_D = ||D - D*||^2_{Sigma_D}, the Mahalanobis norm]:
            depth_loss = depth_mult * mean_r( [z_r > 0] * (D_r - z_r)^2 / Sigma_r ),
        a variance that is not positive and finite drops the ray's depth term; Sigma = 1 is the plain L2 term."""
        d = {"rgb_loss": self.rgb_mult * torch.mean((rgb - gt_rgb) ** 2)}
        if gt_depth is not None:
            z = gt_depth * directions_norm
            mask = (z > 0).double()
            if gt_depth_cov is not None:
                var = gt_depth_cov.double()
                ok = (var > 0) & torch.isfinite(var)
                mask = mask * torch.where(ok, 1.0 / torch.where(ok, var, torch.ones_like(var)), torch.zeros_like(var))
            d["depth_loss"] = self.depth_mult * torch.mean(((depth - z) ** 2) * mask)
        return d


def alive_counts(counts, dt, density_pre, min_transmittance=1e-4):
    """numpy restatement of nvo_ngp_count_alive [UPSTREAM instant-ngp compute_loss_kernel_train_nerf: the loop over a
    ray's samples opens with `if (T < EPSILON) break`, EPSILON = 1e-4]: kept[r] = index of the first sample ray r reaches
    with a transmittance T_j = exp(-sum_{i<j} min(exp(pre_i) dt_i, 128)) below the threshold (counts[r] when none is).
    counts [R]; dt, density_pre: ray-major lists / [R, max] arrays of the marched samples.  Also returns, per ray, how close
    the deciding transmittances came to the threshold (min |T / thr - 1| over its samples): a ray whose margin is below the
    kernel's fp32 rounding (~1e-5) is a tie that may legitimately fall either way."""
    R = len(counts)
    kept = np.zeros(R, dtype=np.int64)
    margin = np.full(R, np.inf)
    for r in range(R):
        n = int(counts[r])
        if n == 0:
            continue
        dd = np.minimum(np.exp(np.asarray(density_pre[r][:n], dtype=np.float64)) * np.asarray(dt[r][:n], dtype=np.float64), 128.0)
        T = np.exp(-(np.cumsum(dd) - dd))
        below = np.nonzero(T < min_transmittance)[0]
        kept[r] = below[0] if len(below) else n
        margin[r] = np.min(np.abs(T / min_transmittance - 1.0))
    return kept, margin


def compact_offsets(kept, capacity):
    """The packing rule of nvo_occ_pack: exclusive scan of kept[R]; a ray whose samples would pass ``capacity`` is dropped
    (count 0) and keeps its slot range.  Returns (counts, offsets [R+1], total)."""
    kept = np.asarray(kept, dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(kept)])
    counts = np.where(offsets[:-1] + kept > capacity, 0, kept)
    return counts, offsets, int(offsets[-1])
