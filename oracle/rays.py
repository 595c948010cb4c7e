"""CPU oracle: ray-side arithmetic of the nerfacto training step (TEST INFRASTRUCTURE; parity unpinned).

torch-CPU, dtype-generic (float64 in tests), differentiable through autograd.  Every function
restates a piece of nerfstudio (jens-nau fork, un-vendored -- SURVEY.md section 3.3 [UPSTREAM]) that
the reference reaches from trainer.train_iteration (/root/reference/nerf_vo/mapping/nerfstudio.py:151)
with the configuration of /root/reference/nerf_vo/mapping/nerfstudio.py:47-103.
"""
from __future__ import annotations

import math

import torch

EPS = 1.0e-7  # nerfstudio model_components/losses.py EPS


# ------------------------------------------------------------------------------------------------
# cameras / ray generation  (nerfstudio cameras/cameras.py Cameras.generate_rays, PERSPECTIVE, no
# distortion; reference builds the Cameras at nerfstudio_utils.py:90-107)
# ------------------------------------------------------------------------------------------------
def generate_rays(ray_indices, intrinsics, c2w):
    """ray_indices [R,3] int64 (camera, y, x); intrinsics [F,4] (fx,fy,cx,cy); c2w [F,3,4].
    Returns origins [R,3], unit directions [R,3], directions_norm [R,1], pixel_area [R,1]."""
    cam, y, x = ray_indices[:, 0], ray_indices[:, 1], ray_indices[:, 2]
    dt = c2w.dtype
    fx, fy, cx, cy = (intrinsics[cam, i].to(dt) for i in range(4))
    px = x.to(dt) + 0.5
    py = y.to(dt) + 0.5

    def cam_dir(px_, py_):
        return torch.stack([(px_ - cx) / fx, -(py_ - cy) / fy, -torch.ones_like(px_)], dim=-1)

    rot = c2w[cam, :, :3]

    def world(d):
        return torch.einsum("rij,rj->ri", rot, d)

    d0 = world(cam_dir(px, py))
    dxv = world(cam_dir(px + 1.0, py))
    dyv = world(cam_dir(px, py + 1.0))
    norm = d0.norm(dim=-1, keepdim=True)
    d0n = d0 / norm
    dxn = dxv / dxv.norm(dim=-1, keepdim=True)
    dyn = dyv / dyv.norm(dim=-1, keepdim=True)
    dx = torch.sqrt(((d0n - dxn) ** 2).sum(-1))
    dy = torch.sqrt(((d0n - dyn) ** 2).sum(-1))
    pixel_area = (dx * dy)[:, None]
    origins = c2w[cam, :, 3]
    return origins, d0n, norm, pixel_area


# ------------------------------------------------------------------------------------------------
# SE3 camera optimizer (nerfstudio cameras/lie_groups.py exp_map_SE3; CameraOptimizerConfig(mode=
# 'SE3') at /root/reference/nerf_vo/mapping/nerfstudio.py:64)
# ------------------------------------------------------------------------------------------------
def exp_map_se3(tangent):
    """tangent [N,6] (translation | rotation) -> [N,3,4]."""
    lin = tangent[:, :3].reshape(-1, 3, 1)
    ang = tangent[:, 3:].reshape(-1, 3, 1)
    theta = torch.linalg.norm(ang, dim=1).unsqueeze(1)
    theta2 = theta ** 2
    theta3 = theta ** 3
    near_zero = theta < 1e-2
    one = torch.ones(1, dtype=tangent.dtype)
    theta_nz = torch.where(near_zero, one, theta)
    theta2_nz = torch.where(near_zero, one, theta2)
    theta3_nz = torch.where(near_zero, one, theta3)
    sine = theta.sin()
    cosine = torch.where(near_zero, 8 / (4 + theta2) - 1, theta.cos())
    sine_by_theta = torch.where(near_zero, 0.5 * cosine + 0.5, sine / theta_nz)
    omc_by_theta2 = torch.where(near_zero, 0.5 * sine_by_theta, (1 - cosine) / theta2_nz)
    n = tangent.shape[0]
    rot = omc_by_theta2 * ang @ ang.transpose(1, 2)
    rot = rot + cosine.view(-1, 1, 1) * torch.eye(3, dtype=tangent.dtype)
    temp = sine_by_theta.view(-1, 1) * ang.view(-1, 3)
    skew = torch.zeros(n, 3, 3, dtype=tangent.dtype)
    skew[:, 0, 1] = -temp[:, 2]
    skew[:, 1, 0] = temp[:, 2]
    skew[:, 0, 2] = temp[:, 1]
    skew[:, 2, 0] = -temp[:, 1]
    skew[:, 1, 2] = -temp[:, 0]
    skew[:, 2, 1] = temp[:, 0]
    rot = rot + skew
    sine_by_theta_t = torch.where(near_zero, 1 - theta2 / 6, sine_by_theta)
    omc_t = torch.where(near_zero, 0.5 - theta2 / 24, omc_by_theta2)
    tms = torch.where(near_zero, 1.0 / 6 - theta2 / 120, (theta - sine) / theta3_nz)
    trans = sine_by_theta_t * lin
    trans = trans + omc_t * torch.cross(ang, lin, dim=1)
    trans = trans + tms * (ang @ (ang.transpose(1, 2) @ lin))
    return torch.cat([rot, trans], dim=2)


def exp_map_so3xr3(tangent):
    """nerfstudio cameras/lie_groups.py exp_map_SO3xR3: Rodrigues rotation (|w|^2 clamped at 1e-4) and
    the translation taken directly from the tangent's first three entries."""
    log_rot = tangent[:, 3:]
    nrms = (log_rot * log_rot).sum(1)
    rot_angles = torch.clamp(nrms, 1e-4).sqrt()
    inv = 1.0 / rot_angles
    fac1 = inv * rot_angles.sin()
    fac2 = inv * inv * (1.0 - rot_angles.cos())
    n = tangent.shape[0]
    zeros = torch.zeros(n, dtype=tangent.dtype)
    skews = torch.stack([
        torch.stack([zeros, -log_rot[:, 2], log_rot[:, 1]], dim=1),
        torch.stack([log_rot[:, 2], zeros, -log_rot[:, 0]], dim=1),
        torch.stack([-log_rot[:, 1], log_rot[:, 0], zeros], dim=1)], dim=1)
    rot = fac1[:, None, None] * skews + fac2[:, None, None] * (skews @ skews) + torch.eye(3, dtype=tangent.dtype)[None]
    return torch.cat([rot, tangent[:, :3, None]], dim=2)


def camera_opt_regularizer(pose_adjustment, trans_l2_penalty=1e-2, rot_l2_penalty=1e-3):
    """CameraOptimizer.get_loss_dict (nerfstudio >= 1.0 [UPSTREAM])."""
    return (pose_adjustment[:, :3].norm(dim=-1).mean() * trans_l2_penalty
            + pose_adjustment[:, 3:].norm(dim=-1).mean() * rot_l2_penalty)


def apply_pose_correction(origins, directions, corrections):
    """CameraOptimizer.apply_to_raybundle: o += t, d = R d (corrections [R,3,4] already gathered)."""
    return origins + corrections[:, :3, 3], torch.einsum("rij,rj->ri", corrections[:, :3, :3], directions)


def pose_multiply(a, b):
    """nerfstudio utils/poses.multiply for [.,3,4] poses (used at nerfstudio.py:208)."""
    r = a[..., :3, :3] @ b[..., :3, :3]
    t = a[..., :3, 3:] + a[..., :3, :3] @ b[..., :3, 3:]
    return torch.cat([r, t], dim=-1)


# ------------------------------------------------------------------------------------------------
# samplers (nerfstudio model_components/ray_samplers.py: UniformLinDispPiecewiseSampler, PDFSampler)
# ------------------------------------------------------------------------------------------------
def spacing_fn(x):
    return torch.where(x < 1, x / 2, 1 - 1 / (2 * x))


def spacing_fn_inv(x):
    return torch.where(x < 0.5, 2 * x, 1 / (2 - 2 * x))


def spacing_to_euclidean(s, near, far):
    s_near = spacing_fn(torch.as_tensor(near, dtype=s.dtype))
    s_far = spacing_fn(torch.as_tensor(far, dtype=s.dtype))
    return spacing_fn_inv(s * s_far + (1 - s) * s_near)


def sample_uniform_lindisp(num_rays, num_samples, near, far, jitter=None, dtype=torch.float64):
    """Piecewise uniform / linear-in-disparity bins.  jitter [R,1] in [0,1) (single_jitter) or None
    for the un-stratified (eval) bins.  Returns (sbins [R,S+1], tbins [R,S+1])."""
    bins = torch.linspace(0.0, 1.0, num_samples + 1, dtype=dtype)[None, :].expand(num_rays, -1)
    if jitter is not None:
        centers = (bins[..., 1:] + bins[..., :-1]) / 2.0
        upper = torch.cat([centers, bins[..., -1:]], -1)
        lower = torch.cat([bins[..., :1], centers], -1)
        bins = lower + (upper - lower) * jitter.to(dtype)
    return bins, spacing_to_euclidean(bins, near, far)


def sample_pdf(sbins, weights, num_samples, near, far, jitter=None, histogram_padding=0.01, eps=1e-5):
    """PDFSampler (include_original=False, single_jitter).  sbins [R,S_in+1] (spacing domain),
    weights [R,S_in] (already annealed).  Returns detached (sbins_out, tbins_out) [R,num_samples+1]."""
    dtype = weights.dtype
    num_bins = num_samples + 1
    w = weights + histogram_padding
    w_sum = torch.sum(w, dim=-1, keepdim=True)
    padding = torch.relu(eps - w_sum)
    w = w + padding / w.shape[-1]
    w_sum = w_sum + padding
    pdf = w / w_sum
    cdf = torch.min(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    u = torch.linspace(0.0, 1.0 - (1.0 / num_bins), steps=num_bins, dtype=dtype)
    u = u.expand(*cdf.shape[:-1], num_bins)
    if jitter is not None:
        u = u + jitter.to(dtype) / num_bins
    else:
        u = u + 1.0 / (2 * num_bins)
    u = u.contiguous()
    existing = sbins
    inds = torch.searchsorted(cdf.contiguous(), u, side="right")
    below = torch.clamp(inds - 1, 0, existing.shape[-1] - 1)
    above = torch.clamp(inds, 0, existing.shape[-1] - 1)
    cdf_g0 = torch.gather(cdf, -1, below)
    bins_g0 = torch.gather(existing, -1, below)
    cdf_g1 = torch.gather(cdf, -1, above)
    bins_g1 = torch.gather(existing, -1, above)
    t = torch.clip(torch.nan_to_num((u - cdf_g0) / (cdf_g1 - cdf_g0), 0), 0, 1)
    bins = (bins_g0 + t * (bins_g1 - bins_g0)).detach()
    return bins, spacing_to_euclidean(bins, near, far)


# ------------------------------------------------------------------------------------------------
# field glue (nerfstudio field_components/spatial_distortions.py SceneContraction(order=inf),
# fields/nerfacto_field.py get_density, field_components/activations.py trunc_exp)
# ------------------------------------------------------------------------------------------------
def contract_linf(x):
    mag = x.abs().amax(dim=-1, keepdim=True)
    safe = torch.where(mag < 1, torch.ones_like(mag), mag)
    return torch.where(mag < 1, x, (2 - (1 / safe)) * (x / safe))


def sample_positions(origins, directions, tbins):
    """Frustums.get_positions: o + d * (start + end) / 2  -> [R,S,3]."""
    mid = (tbins[:, :-1] + tbins[:, 1:]) / 2
    return origins[:, None, :] + directions[:, None, :] * mid[:, :, None]


def normalized_positions(positions):
    """contract -> (x+2)/4 -> selector mask; returns (x01 [.,3] with masked rows zeroed, selector)."""
    x = (contract_linf(positions) + 2.0) / 4.0
    selector = ((x > 0.0) & (x < 1.0)).all(dim=-1)
    return x * selector[..., None], selector


class _TruncExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(torch.clamp(x, min=-15, max=15))


trunc_exp = _TruncExp.apply


def get_weights(tbins, densities):
    """RaySamples.get_weights.  tbins [R,S+1], densities [R,S] -> weights [R,S]."""
    deltas = tbins[:, 1:] - tbins[:, :-1]
    delta_density = deltas * densities
    alphas = 1 - torch.exp(-delta_density)
    transmittance = torch.cumsum(delta_density[..., :-1], dim=-1)
    transmittance = torch.cat([torch.zeros_like(transmittance[..., :1]), transmittance], dim=-1)
    transmittance = torch.exp(-transmittance)
    return torch.nan_to_num(alphas * transmittance)


# ------------------------------------------------------------------------------------------------
# renderers (nerfstudio model_components/renderers.py)
# ------------------------------------------------------------------------------------------------
def render_rgb_last_sample(weights, rgb):
    """RGBRenderer(background_color='last_sample') in training mode (no clamp)."""
    comp = torch.sum(weights[..., None] * rgb, dim=-2)
    acc = torch.sum(weights, dim=-1, keepdim=True)
    return comp + rgb[..., -1, :] * (1.0 - acc)


def render_accumulation(weights):
    return torch.sum(weights, dim=-1, keepdim=True)


def render_depth_median(weights, tbins):
    steps = (tbins[:, :-1] + tbins[:, 1:]) / 2
    cum = torch.cumsum(weights, dim=-1)
    split = torch.full((weights.shape[0], 1), 0.5, dtype=weights.dtype)
    idx = torch.searchsorted(cum.contiguous(), split, side="left")
    idx = torch.clamp(idx, 0, steps.shape[-1] - 1)
    return torch.gather(steps, dim=-1, index=idx)


def render_depth_expected(weights, tbins):
    eps = 1e-10
    steps = (tbins[:, :-1] + tbins[:, 1:]) / 2
    depth = torch.sum(weights * steps, dim=-1, keepdim=True) / (torch.sum(weights, -1, keepdim=True) + eps)
    return torch.clip(depth, steps.min(), steps.max())


# ------------------------------------------------------------------------------------------------
# losses (nerfstudio model_components/losses.py; imported by the reference at
# /root/reference/nerf_vo/mapping/nerfstudio_utils.py:25)
# ------------------------------------------------------------------------------------------------
def _outer(t0_starts, t0_ends, t1_starts, t1_ends, y1):
    cy1 = torch.cat([torch.zeros_like(y1[..., :1]), torch.cumsum(y1, dim=-1)], dim=-1)
    idx_lo = torch.searchsorted(t1_starts.contiguous(), t0_starts.contiguous(), side="right") - 1
    idx_lo = torch.clamp(idx_lo, min=0, max=y1.shape[-1] - 1)
    idx_hi = torch.searchsorted(t1_ends.contiguous(), t0_ends.contiguous(), side="right")
    idx_hi = torch.clamp(idx_hi, min=0, max=y1.shape[-1] - 1)
    cy1_lo = torch.take_along_dim(cy1[..., :-1], idx_lo, dim=-1)
    cy1_hi = torch.take_along_dim(cy1[..., 1:], idx_hi, dim=-1)
    return cy1_hi - cy1_lo


def lossfun_outer(t, w, t_env, w_env):
    w_outer = _outer(t[..., :-1], t[..., 1:], t_env[..., :-1], t_env[..., 1:], w_env)
    return torch.clip(w - w_outer, min=0) ** 2 / (w + EPS)


def interlevel_loss(weights_list, sbins_list):
    """mip-NeRF 360 proposal loss; main-level histogram is detached."""
    c = sbins_list[-1].detach()
    w = weights_list[-1].detach()
    loss = 0.0
    for sb, wp in zip(sbins_list[:-1], weights_list[:-1]):
        loss = loss + torch.mean(lossfun_outer(c, w, sb, wp))
    return loss


def lossfun_distortion(t, w):
    ut = (t[..., 1:] + t[..., :-1]) / 2
    dut = torch.abs(ut[..., :, None] - ut[..., None, :])
    loss_inter = torch.sum(w * torch.sum(w[..., None, :] * dut, dim=-1), dim=-1)
    loss_intra = torch.sum(w ** 2 * (t[..., 1:] - t[..., :-1]), dim=-1) / 3
    return loss_inter + loss_intra


def distortion_loss(weights_main, sbins_main):
    return torch.mean(lossfun_distortion(sbins_main, weights_main))


def ds_nerf_depth_loss(weights, tbins, termination_depth, sigma):
    """weights [R,S], tbins [R,S+1], termination_depth [R,1] (already x directions_norm)."""
    steps = (tbins[:, :-1] + tbins[:, 1:]) / 2
    lengths = tbins[:, 1:] - tbins[:, :-1]
    mask = (termination_depth > 0).to(weights.dtype)
    loss = -torch.log(weights + EPS) * torch.exp(-((steps - termination_depth) ** 2) / (2 * sigma)) * lengths
    loss = loss.sum(-1, keepdim=True) * mask
    return torch.mean(loss)


def render_normals_shaded(weights, sample_normals):
    """NormalsRenderer (sum_i w_i n_i, then safe_normalize = v / (|v| + 1e-10)) followed by NormalsShader
    ((n + 1) / 2) -- what nerfacto stores in outputs['normals'] [UPSTREAM] and what the reference's
    normal-loss hook compares with the (n+1)/2-mapped targets of DynamicDataset.get_dataset
    (ref: nerf_vo/mapping/nerfstudio_utils.py:145-153, 337-350)."""
    n = torch.sum(weights[..., None] * sample_normals, dim=-2)
    n = n / (torch.linalg.norm(n, dim=-1, keepdim=True) + 1e-10)
    return (n + 1.0) / 2.0


def monosdf_normal_loss(normal_pred, normal_gt):
    normal_gt = torch.nn.functional.normalize(normal_gt, p=2, dim=-1)
    normal_pred = torch.nn.functional.normalize(normal_pred, p=2, dim=-1)
    l1 = torch.abs(normal_pred - normal_gt).sum(dim=-1).mean()
    cos = (1.0 - torch.sum(normal_pred * normal_gt, dim=-1)).mean()
    return l1 + cos


def proposal_anneal(step: int, max_iters: int = 1000, slope: float = 10.0) -> float:
    """nerfacto set_anneal callback (mip-NeRF 360 eq. 18)."""
    frac = min(max(step / max_iters, 0.0), 1.0)
    return slope * frac / ((slope - 1) * frac + 1)


def proposal_update_due(step: int, steps_since_update: int, warmup: int = 5000, every: int = 5) -> bool:
    """ProposalNetworkSampler: proposal nets run with gradients when this is True."""
    sched = min(max(step / warmup * every if warmup > 0 else every, 1.0), float(every))
    return steps_since_update > sched or step < 10


def psnr_reference(img_a, img_b):
    """calculate_psnr + calculate_psnr_color of /root/reference/evaluation/evaluation_utils.py:289-318
    INCLUDING its uint8 wrap-around: (a - b) ** 2 is evaluated in uint8."""
    import numpy as np

    a = np.asarray(img_a)
    b = np.asarray(img_b)
    assert a.dtype == np.uint8 and b.dtype == np.uint8
    vals = []
    for c in range(3):
        with np.errstate(over="ignore"):
            mse = np.mean((a[..., c] - b[..., c]) ** 2)
        vals.append(float("inf") if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse)))
    return sum(vals) / 3.0


def psnr_float(img_a, img_b):
    """Conventional PSNR on the same uint8 images, per channel then averaged (no wrap-around)."""
    import numpy as np

    a = np.asarray(img_a).astype(np.float64)
    b = np.asarray(img_b).astype(np.float64)
    vals = []
    for c in range(3):
        mse = np.mean((a[..., c] - b[..., c]) ** 2)
        vals.append(float("inf") if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse)))
    return sum(vals) / 3.0
