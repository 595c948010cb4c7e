#!/usr/bin/env python3
"""Generates tests/golden/oracle_golden.npz from the CPU oracle (python tests/golden/make_golden.py).

The reference holds no golden vectors for this path and cannot be imported here (SURVEY.md section
8c), so these fixtures are produced by THIS repository's oracle and pin it against regressions; they
do not pin it against the reference ("parity unpinned", oracle/__init__.py).  Inputs are regenerated
from the seeds below; only expected outputs (and small inputs) are stored."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import grid as G  # noqa: E402
from oracle import mlp as M  # noqa: E402
from oracle import rays as Rr  # noqa: E402
from oracle import sh as S  # noqa: E402

GRIDS = {"main": (16, 19, 16, 2048), "prop0": (5, 17, 16, 128), "prop1": (5, 17, 16, 256)}


def pls(b, m, L):
    return float(np.exp((np.log(m) - np.log(b)) / (L - 1)))


def seeded_points(n, seed):
    x = np.random.default_rng(seed).random((n, 3), dtype=np.float32)
    x[0], x[1], x[2] = 0.0, 1.0, 0.5
    return x


def build():
    out = {}
    # G1 level tables, G2 corner indices
    for name, (L, T, b, m) in GRIDS.items():
        spec = G.make_grid_spec(L, 2, T, b, pls(b, m, L))
        out[f"g1_{name}_levels"] = spec.levels
        out[f"g1_{name}_scales"] = spec.scales
        idx, w = G.grid_indices_c(spec, seeded_points(64, 11))
        out[f"g2_{name}_indices"] = idx
        out[f"g2_{name}_weights"] = w
    # G3 encoded features + gradients for a seeded table (prop0 grid: small enough to regenerate fast)
    L, T, b, m = GRIDS["prop0"]
    spec = G.make_grid_spec(L, 2, T, b, pls(b, m, L))
    table = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, (spec.n_entries, 2))).requires_grad_(True)
    x = torch.from_numpy(seeded_points(64, 12)).double().requires_grad_(True)
    y = G.grid_encode(spec, x, table)
    y.square().sum().backward()
    out["g3_features"] = y.detach().numpy()
    out["g3_dx"] = x.grad.numpy()
    out["g3_dtable_sum_per_level"] = np.array([table.grad[int(spec.levels[l, 0]):int(spec.levels[l, 0] + spec.levels[l, 1])].sum().item()
                                               for l in range(L)])
    # G4 MLP forward for each network shape on the path
    for tag, (n_in, n_out, width, n_hidden, act, oact) in {
            "base": (32, 16, 64, 1, "ReLU", "None"), "color": (63, 3, 64, 2, "ReLU", "Sigmoid"),
            "normals": (27, 64, 64, 3, "ReLU", "None"), "prop": (10, 1, 16, 1, "ReLU", "None")}.items():
        g = torch.Generator().manual_seed(21)
        p = (torch.rand(M.mlp_n_params(n_in, n_out, width, n_hidden), generator=g, dtype=torch.float64) * 2 - 1) * 0.3
        p = p.half().double()
        xin = torch.randn(32, n_in, generator=g, dtype=torch.float64)
        out[f"g4_{tag}_out"] = M.mlp_forward(xin, M.split_weights(p, n_in, n_out, width, n_hidden), act, oact).numpy()
    # G5 SH degree 4
    g = torch.Generator().manual_seed(31)
    d = torch.nn.functional.normalize(torch.randn(100, 3, generator=g, dtype=torch.float64), dim=-1)
    out["g5_dirs"] = d.numpy()
    out["g5_sh4"] = S.sh_encode((d + 1) / 2, 4).numpy()
    # G6 samplers
    jit = torch.rand(8, 1, generator=g, dtype=torch.float64)
    sb, tb = Rr.sample_uniform_lindisp(8, 256, 0.05, 1000.0, jit)
    out["g6_jitter"] = jit.numpy()
    out["g6_lindisp_sbins"], out["g6_lindisp_tbins"] = sb.numpy(), tb.numpy()
    w = torch.rand(8, 256, generator=g, dtype=torch.float64) ** 4
    out["g6_pdf_weights"] = w.numpy()
    sb2, tb2 = Rr.sample_pdf(sb, w, 96, 0.05, 1000.0, jit)
    out["g6_pdf_sbins"], out["g6_pdf_tbins"] = sb2.numpy(), tb2.numpy()
    # G7 weights / rendering
    dens = torch.rand(8, 96, generator=g, dtype=torch.float64) * 3
    wts = Rr.get_weights(tb2, dens)
    rgb = torch.rand(8, 96, 3, generator=g, dtype=torch.float64)
    out["g7_density"], out["g7_rgb_samples"] = dens.numpy(), rgb.numpy()
    out["g7_weights"] = wts.numpy()
    out["g7_rgb"] = Rr.render_rgb_last_sample(wts, rgb).numpy()
    out["g7_depth_median"] = Rr.render_depth_median(wts, tb2).numpy()
    out["g7_depth_expected"] = Rr.render_depth_expected(wts, tb2).numpy()
    # G8 losses
    w_main = Rr.get_weights(tb2[:, ::2], dens[:, ::2][:, :48])
    out["g8_interlevel"] = float(Rr.interlevel_loss([wts, w_main], [sb2, sb2[:, ::2]]))
    out["g8_distortion"] = float(Rr.distortion_loss(wts, sb2))
    term = torch.rand(8, 1, generator=g, dtype=torch.float64) * 3
    out["g8_termination"] = term.numpy()
    out["g8_depth"] = float(Rr.ds_nerf_depth_loss(wts, tb2, term, 0.001))
    n1 = torch.randn(16, 3, generator=g, dtype=torch.float64)
    n2 = torch.randn(16, 3, generator=g, dtype=torch.float64)
    out["g8_normal_inputs"] = torch.stack([n1, n2]).numpy()
    out["g8_monosdf"] = float(Rr.monosdf_normal_loss(n1, n2))
    # G9 SE3 exp map, pose composition
    tang = torch.cat([torch.randn(6, 6, generator=g, dtype=torch.float64) * 0.3,
                      torch.randn(2, 6, generator=g, dtype=torch.float64) * 1e-3])
    out["g9_tangent"] = tang.numpy()
    e = Rr.exp_map_se3(tang)
    out["g9_exp"] = e.numpy()
    out["g9_multiply"] = Rr.pose_multiply(e[:4], e[4:]).numpy()
    # G6b occupancy DDA hit lists on a seeded bitfield (bit-exact integer/IEEE path)
    from oracle import occgrid as O
    rng_o = np.random.default_rng(51)
    ogrid = (rng_o.random((3, O.CELLS), dtype=np.float32) ** 6) * 0.08
    obf = O.grid_to_bitfield(ogrid, 3)
    oo = (rng_o.random((16, 3), dtype=np.float32) - 0.5) * 0.8 + 0.5
    od = rng_o.normal(size=(16, 3)).astype(np.float32)
    od /= np.linalg.norm(od, axis=1, keepdims=True)
    ojit = rng_o.random(16).astype(np.float32)
    oc, ot, odt = O.march_rays(oo, od, obf, 3, 1 / 256, 0.0, ojit, max_out=64)
    out["g6b_occ_origins"], out["g6b_occ_dirs"], out["g6b_occ_jitter"] = oo, od, ojit
    out["g6b_occ_counts"], out["g6b_occ_t"], out["g6b_occ_dt"] = oc, ot, odt
    out["g6b_occ_bitfield_popcount"] = np.array([int(np.unpackbits(obf[l]).sum()) for l in range(3)])
    # G10 PSNR (both definitions)
    rng = np.random.default_rng(41)
    a = rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)
    bb = np.clip(a.astype(np.int32) + rng.integers(-30, 31, a.shape), 0, 255).astype(np.uint8)
    out["g10_a"], out["g10_b"] = a, bb
    out["g10_psnr_reference"] = Rr.psnr_reference(a, bb)
    out["g10_psnr_float"] = Rr.psnr_float(a, bb)
    # G7b rendered normals (NormalsRenderer + NormalsShader) and the normal loss on them; own generator so
    # that the vectors above keep their values
    g2 = torch.Generator().manual_seed(77)
    sn = torch.nn.functional.normalize(torch.randn(8, 96, 3, generator=g2, dtype=torch.float64), dim=-1)
    out["g7b_sample_normals"] = sn.numpy()
    shaded = Rr.render_normals_shaded(wts, sn)
    out["g7b_normals_shaded"] = shaded.numpy()
    gtn = (torch.nn.functional.normalize(torch.randn(8, 3, generator=g2, dtype=torch.float64), dim=-1) + 1) / 2
    out["g7b_gt_normal"] = gtn.numpy()
    out["g8b_monosdf_on_shaded"] = float(Rr.monosdf_normal_loss(shaded, gtn))
    return out


if __name__ == "__main__":
    data = build()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_golden.npz")
    np.savez_compressed(path, **data)
    print(f"wrote {path}: {len(data)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")
