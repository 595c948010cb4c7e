"""CPU oracle: the depth-nerfacto training step NeRF-VO's mapping runs (TEST INFRASTRUCTURE; parity
unpinned -- see oracle/__init__.py).

Pure torch-CPU restatement (autograd for every gradient) of one ``trainer.train_iteration``
(/root/reference/nerf_vo/mapping/nerfstudio.py:151) with the model configuration of
/root/reference/nerf_vo/mapping/nerfstudio.py:62-82 and nerfacto's defaults [UPSTREAM, SURVEY.md
section 3.3]: proposal sampling 256 -> 96 -> 48, HashMLPDensityField x2, NerfactoField (hash grid ->
MLP 32-64-16 -> trunc_exp density; SH4 | geo15 | appearance32 -> MLP 64-64-3 sigmoid), last-sample
background compositing, rgb MSE + interlevel + distortion + DS-NeRF depth losses, Adam.

It is also the "reference CPU PyTorch path" timed as ``cpu_baseline`` (kind "port") by bench.py
(BASELINE.json configs[0]).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import torch

from . import grid as G
from . import mlp as M
from . import rays as Rr
from . import sh as S


def _pls(base, max_res, n_levels):
    return float(np.exp((np.log(max_res) - np.log(base)) / (n_levels - 1)))


@dataclass
class OracleConfig:
    num_images: int = 192
    near_plane: float = 0.05
    far_plane: float = 1000.0
    num_proposal_samples: tuple = (256, 96)
    num_nerf_samples: int = 48
    main_grid: tuple = (16, 19, 16, 2048)           # n_levels, log2_T, base_res, max_res
    proposal_grids: tuple = ((5, 17, 16, 128), (5, 17, 16, 256))
    density_bias: float = -1.0
    histogram_padding: float = 0.01
    rgb_loss_mult: float = 1.0
    interlevel_loss_mult: float = 1.0
    distortion_loss_mult: float = 0.002
    depth_loss_mult: float = 0.001
    depth_sigma: float = 0.001
    normal_loss_mult: float = 5e-6          # ref: nerf_vo/mapping/nerfstudio.py:77
    emulate_fp16: bool = True
    dtype: torch.dtype = torch.float64


from .quant import q16 as _q16  # 16-bit storage emulation in the ACTIVE format (fp16 | bf16), see quant.py


class NerfactoOracle:
    """Parameters are a dict of leaf tensors (fp16-representable values when emulate_fp16):
    base_mlp, base_grid [E,2], color_mlp, embedding [F,32], prop{k}_mlp, prop{k}_grid."""

    def __init__(self, cfg: OracleConfig):
        self.cfg = cfg
        L, T, b, m = cfg.main_grid
        self.main_spec = G.make_grid_spec(L, 2, T, b, _pls(b, m, L))
        self.prop_specs = [G.make_grid_spec(L, 2, T, b, _pls(b, m, L)) for (L, T, b, m) in cfg.proposal_grids]
        self.params: dict[str, torch.Tensor] = {}

    # -------------------------------------------------------------------------------------------
    def param_shapes(self) -> dict:
        cfg = self.cfg
        shapes = {
            "base_mlp": (M.mlp_n_params(self.main_spec.n_output_dims, 16, 64, 1),),
            "base_grid": (self.main_spec.n_entries, 2),
            "color_mlp": (M.mlp_n_params(63, 3, 64, 2),),
            "embedding": (cfg.num_images, 32),
        }
        for k, sp in enumerate(self.prop_specs):
            shapes[f"prop{k}_mlp"] = (M.mlp_n_params(sp.n_output_dims, 1, 16, 1),)
            shapes[f"prop{k}_grid"] = (sp.n_entries, 2)
        return shapes

    def init_random(self, seed=0, grid_scale=1e-4):
        g = torch.Generator().manual_seed(seed)
        dt = self.cfg.dtype
        for name, shape in self.param_shapes().items():
            if name.endswith("_grid"):
                p = (torch.rand(shape, generator=g) * 2 - 1) * grid_scale
            elif name == "embedding":
                p = torch.randn(shape, generator=g)
            else:
                width = 16 if name.startswith("prop") else 64
                p = (torch.rand(shape, generator=g) * 2 - 1) * float(np.sqrt(6.0 / (2 * width)))
            self.params[name] = p.to(dt).requires_grad_(True)

    # -------------------------------------------------------------------------------------------
    def _density(self, spec, grid, mlp_flat, width, n_out, origins, directions, tbins, want_normals=False):
        cfg = self.cfg
        pos = Rr.sample_positions(origins, directions, tbins)
        x01, selector = Rr.normalized_positions(pos)
        if want_normals and not x01.requires_grad:
            x01 = x01.detach().requires_grad_(True)   # NerfactoField.get_density: _sample_locations.requires_grad = True
        flat = x01.reshape(-1, 3)
        enc = G.grid_encode(spec, flat, grid, quantize_output=cfg.emulate_fp16)
        ws = M.split_weights(mlp_flat, spec.n_output_dims, n_out, width, 1)
        out = M.mlp_forward(enc, ws, "ReLU", "None", pad_value=0.0, emulate_fp16=cfg.emulate_fp16)
        pre = out[:, 0].reshape(tbins.shape[0], -1)
        density = Rr.trunc_exp(pre + cfg.density_bias) * selector.to(pre.dtype)
        if want_normals:
            # NerfactoField.get_normals [UPSTREAM]: autograd.grad(density_before_activation, sample_locations,
            # grad_outputs=ones, retain_graph=True) -- no create_graph, the normals are constants
            (gx,) = torch.autograd.grad(pre, x01, grad_outputs=torch.ones_like(pre), retain_graph=True)
            normals = -torch.nn.functional.normalize(gx, dim=-1)
            return density, out, normals.detach()
        return density, out

    def forward(self, origins, directions, directions_norm, cam_idx, jitters, anneal=1.0, training=True,
                normals=False, sample_normals_override=None):
        """Returns dict with per-level (sbins, tbins, weights), rgb per sample, rendered outputs."""
        cfg = self.cfg
        P = self.params
        dt = cfg.dtype
        R = origins.shape[0]
        j = jitters if jitters is not None else (None, None, None)
        sb, tb = Rr.sample_uniform_lindisp(R, cfg.num_proposal_samples[0], cfg.near_plane, cfg.far_plane,
                                           None if j[0] is None else j[0].reshape(R, 1), dtype=dt)
        sbins_list, tbins_list, weights_list = [], [], []
        n_next = (*cfg.num_proposal_samples[1:], cfg.num_nerf_samples)
        for k, spec in enumerate(self.prop_specs):
            dens, _ = self._density(spec, P[f"prop{k}_grid"], P[f"prop{k}_mlp"], 16, 1, origins, directions, tb)
            w = Rr.get_weights(tb, dens)
            sbins_list.append(sb)
            tbins_list.append(tb)
            weights_list.append(w)
            annealed = torch.pow(w.detach(), anneal)
            sb, tb = Rr.sample_pdf(sb, annealed, n_next[k], cfg.near_plane, cfg.far_plane,
                                   None if j[k + 1] is None else j[k + 1].reshape(R, 1), cfg.histogram_padding)
        res = self._density(self.main_spec, P["base_grid"], P["base_mlp"], 64, 16, origins, directions, tb,
                            want_normals=normals)
        dens, base_out = res[0], res[1]
        Sm = cfg.num_nerf_samples
        geo = base_out[:, 1:16]
        d01 = (directions + 1.0) / 2.0
        sh = S.sh_encode(d01, 4)
        if cfg.emulate_fp16:
            sh = _q16(sh)
        sh = sh[:, None, :].expand(R, Sm, 16).reshape(-1, 16)
        if training:
            emb = P["embedding"][cam_idx.long()]
        else:
            emb = P["embedding"].mean(dim=0, keepdim=True)
            if cfg.emulate_fp16:
                emb = _q16(emb)
            emb = emb.expand(R, 32)
        emb = emb[:, None, :].expand(R, Sm, 32).reshape(-1, 32)
        cin = torch.cat([sh, geo, emb], dim=-1)
        cws = M.split_weights(P["color_mlp"], 63, 3, 64, 2)
        rgb = M.mlp_forward(cin, cws, "ReLU", "Sigmoid", pad_value=1.0, emulate_fp16=cfg.emulate_fp16)[:, :3]
        rgb = rgb.reshape(R, Sm, 3)
        w = Rr.get_weights(tb, dens)
        sbins_list.append(sb)
        tbins_list.append(tb)
        weights_list.append(w)
        out_rgb = Rr.render_rgb_last_sample(w, rgb)
        with torch.no_grad():
            depth = Rr.render_depth_median(w, tb)
            acc = Rr.render_accumulation(w)
            steps = (tb[:, :-1] + tb[:, 1:]) / 2
            expected = torch.sum(w * steps, dim=-1, keepdim=True) / (acc + 1e-10)
        if not training:
            out_rgb = out_rgb.clamp(0.0, 1.0)
        out = {"rgb": out_rgb, "depth": depth, "accumulation": acc, "expected_depth": expected,
               "weights_list": weights_list, "sbins_list": sbins_list, "tbins_list": tbins_list,
               "rgb_samples": rgb, "base_out": base_out, "directions_norm": directions_norm}
        if normals:
            # sample_normals_override: tests inject the kernel's own per-sample normals to check the
            # render + loss arithmetic independently of the fp16 backward that produced them
            sn = res[2] if sample_normals_override is None else sample_normals_override
            out["sample_normals"] = res[2]
            out["normals"] = Rr.render_normals_shaded(w, sn)
        return out

    def loss_dict(self, outputs, gt_rgb, gt_depth, gt_normal=None):
        cfg = self.cfg
        wl, sl, tl = outputs["weights_list"], outputs["sbins_list"], outputs["tbins_list"]
        d = {"rgb_loss": cfg.rgb_loss_mult * torch.mean((outputs["rgb"] - gt_rgb) ** 2)}
        d["interlevel_loss"] = cfg.interlevel_loss_mult * Rr.interlevel_loss(wl, sl)
        d["distortion_loss"] = cfg.distortion_loss_mult * Rr.distortion_loss(wl[-1], sl[-1])
        if gt_depth is not None and cfg.depth_loss_mult > 0:
            term = gt_depth.reshape(-1, 1) * outputs["directions_norm"].reshape(-1, 1)
            dl = 0.0
            for w, tb in zip(wl, tl):
                dl = dl + Rr.ds_nerf_depth_loss(w, tb, term, cfg.depth_sigma) / len(wl)
            d["depth_loss"] = cfg.depth_loss_mult * dl
        if gt_normal is not None and cfg.normal_loss_mult > 0 and "normals" in outputs:
            # reference hook: nerf_vo/mapping/nerfstudio_utils.py:337-350
            d["normal_loss"] = cfg.normal_loss_mult * Rr.monosdf_normal_loss(outputs["normals"], gt_normal)
        return d

    def zero_grad(self):
        for p in self.params.values():
            p.grad = None


def adam_reference(p, g, m, v, lr, step, betas=(0.9, 0.999), eps=1e-15):
    """torch.optim.Adam single-tensor update (no amsgrad / weight decay), returns new (p, m, v)."""
    b1, b2 = betas
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / (bc2 ** 0.5) + eps
    return p - (lr / bc1) * (m / denom), m, v
