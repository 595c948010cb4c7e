"""The PSNR half of the metric, like for like: the HIP engine (fp16 operands, fp32 accumulate) and the CPU oracle are
trained from identical initial parameters on identical injected rays and jitters, with the same Adam, and then render
the same held-out and training views.  This is the only honest stand-in available here for "PSNR within 0.2 dB of the
CUDA reference" (BASELINE.json north_star; both PSNR definitions of /root/reference/evaluation/evaluation_utils.py:
289-318 are pinned separately in tests/test_mapping_gpu.py).

The oracle runs in float64 WITH the 16-bit storage points of tiny-cuda-nn emulated (weights, encoded features, hidden
activations and outputs rounded to fp16, exact arithmetic everywhere else) -- what the reference's fp16 tcnn path
computes, minus its fp16 accumulation.  Against that: held-out PSNR within 0.01-0.1 dB, training views within 0.05 dB
(mean over four checkpoints of the last 75 steps; two chaotic trajectories are compared and a single end point of the
same build scatters by +-0.25 dB).  Against PURE float64 (NVO_TEST_ORACLE_FP16=0, informational) the 16-bit format itself
costs about 0.25 dB on the training views at this stage of this small run (20.57-20.68 vs 20.83 dB whatever the loss
scale, 128 ... 65536, or the backward variant) and under 0.11 dB on the held-out views.

Sizes are reduced so that the oracle finishes in about a minute (main grid 16 levels x 2^14, 128 rays, 300 steps);
the kernels are the production ones (same MLP shapes, same samplers, same losses, same fused Adam)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_IMG, N_HELD, H, W, RAYS, STEPS = 6, 2, 48, 64, 128, 300
CHECKPOINTS = {225, 250, 275, 300}


def _oracle_like(eng, ocfg):
    from oracle import mlp as omlp
    from oracle.nerfacto import NerfactoOracle

    orc = NerfactoOracle(ocfg)
    p = eng.params.detach().double().cpu()  # the fp32 master weights

    def seg(name):
        o, s, _ = eng.segments[name]
        return p[o:o + s]

    nb = omlp.mlp_n_params(32, 16, 64, 1)
    npk = omlp.mlp_n_params(10, 1, 16, 1)
    orc.params = {"base_mlp": seg("field.base")[:nb].clone(), "base_grid": seg("field.base")[nb:].clone().view(-1, 2),
                  "color_mlp": seg("field.color").clone(), "embedding": seg("field.embedding").clone().view(N_IMG, 32)}
    for k in range(2):
        orc.params[f"prop{k}_mlp"] = seg(f"proposal.{k}")[:npk].clone()
        orc.params[f"prop{k}_grid"] = seg(f"proposal.{k}")[npk:].clone().view(-1, 2)
    for v in orc.params.values():
        v.requires_grad_(True)
    return orc


def _psnr(pred, gt):
    return float(-10.0 * torch.log10(torch.mean((pred.double().cpu() - gt.double().cpu()) ** 2)))


def test_hip_fp16_training_matches_oracle_psnr(device):
    from nerf_vo_amd.engine import EngineConfig, GridConfig, NerfactoEngine
    from nerf_vo_amd.mapping.dataset import DynamicDataset, opencv_to_opengl
    from nerf_vo_amd.synthetic import make_sequence
    from oracle import rays as Rr
    from oracle.nerfacto import OracleConfig

    torch.manual_seed(11)
    # the oracle runs on the host: a many-core box oversubscribes itself at torch's default thread count
    # (256 threads measured 40x slower than 16 on the bench box)
    saved_threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    seq = make_sequence(N_IMG + N_HELD, H, W, device=device)  # the last views are held out
    ds = DynamicDataset(num_frames=N_IMG + N_HELD, frame_height=H, frame_width=W, device=device, use_normals=False)
    ds.update({"keyframe_indices": torch.arange(N_IMG + N_HELD), "camera_intrinsics": seq["camera_intrinsics"],
               "camera_extrinsics": opencv_to_opengl(seq["camera_extrinsics"]), "frames_color": seq["frames_color"],
               "frames_depth": seq["frames_depth"]})
    intr = ds.camera_intrinsics
    c2w = ds.camera_extrinsics[:, :3, :4].contiguous()
    images, depths = ds.frames_color, ds.frames_depth

    grids = dict(main=(16, 14, 16, 512), props=((5, 12, 16, 64), (5, 12, 16, 128)))
    import json

    extra = json.loads(os.environ.get("NVO_TEST_ENGINE_OPTS", "{}"))  # (A/B of engine options against the same oracle run)
    eng = NerfactoEngine(EngineConfig(num_images=N_IMG, num_rays=RAYS, main_grid=GridConfig(*grids["main"]),
                                      proposal_grids=tuple(GridConfig(*g) for g in grids["props"]), **extra), device)
    orc = _oracle_like(eng, OracleConfig(num_images=N_IMG, main_grid=grids["main"], proposal_grids=grids["props"],
                                         density_bias=eng.cfg.density_bias,
                                         emulate_fp16=os.environ.get("NVO_TEST_ORACLE_FP16", "1") == "1"))
    cfg = eng.cfg
    opt_fields = torch.optim.Adam([orc.params[k] for k in ("base_mlp", "base_grid", "color_mlp", "embedding")],
                                  lr=cfg.lr_fields, betas=cfg.adam_betas, eps=cfg.adam_eps)
    opt_prop = torch.optim.Adam([orc.params[k] for k in ("prop0_mlp", "prop0_grid", "prop1_mlp", "prop1_grid")],
                                lr=cfg.lr_proposal, betas=cfg.adam_betas, eps=cfg.adam_eps)

    g = torch.Generator().manual_seed(5)
    images_c, depths_c = images.cpu().double(), depths.cpu().double()
    intr_c, c2w_c = intr.cpu().double(), c2w.cpu().double()
    hip_loss, orc_loss = [], []

    def render_both(view_ids, stride):
        cams, ys, xs = torch.meshgrid(torch.tensor(view_ids), torch.arange(0, H, stride), torch.arange(0, W, stride),
                                      indexing="ij")
        idx = torch.stack([cams, ys, xs], dim=-1).reshape(-1, 3)
        ro, rd, rn, _ = Rr.generate_rays(idx, intr_c, c2w_c)
        gt = images_c[idx[:, 0], idx[:, 1], idx[:, 2]]
        hip = torch.cat([eng.render_rays(ro[c:c + 2048].float().to(device), rd[c:c + 2048].float().to(device),
                                         rn.reshape(-1)[c:c + 2048].float().to(device))["rgb"].cpu()
                         for c in range(0, idx.shape[0], 2048)])
        with torch.no_grad():
            ref = orc.forward(ro, rd, rn.reshape(-1), idx[:, 0].clamp(max=N_IMG - 1), None, anneal=1.0, training=False)["rgb"]
        return hip, ref, gt

    def evaluate():
        """(held-out HIP, held-out oracle, training-view HIP, training-view oracle, HIP-vs-oracle) PSNR at this point:
        every pixel of the views neither run has seen, every 2nd pixel of the training views."""
        hip, ref, gt = render_both(list(range(N_IMG, N_IMG + N_HELD)), 1)
        hip_t, ref_t, gt_t = render_both(list(range(N_IMG)), 2)
        return _psnr(hip, gt), _psnr(ref, gt), _psnr(hip_t, gt_t), _psnr(ref_t, gt_t), _psnr(hip, ref)

    checkpoints = []
    for step in range(STEPS):
        idx = torch.stack([torch.randint(0, N_IMG, (RAYS,), generator=g), torch.randint(0, H, (RAYS,), generator=g),
                           torch.randint(0, W, (RAYS,), generator=g)], dim=1)
        jit = tuple(torch.rand(RAYS, generator=g) for _ in range(3))
        anneal = eng.anneal_at(eng.step)
        updated = eng.train_step(idx.to(device), intr, c2w, images, depths, jitters=tuple(j.to(device) for j in jit))
        hip_loss.append(eng.loss_dict()["rgb_loss"])
        # the same step in exact arithmetic
        ro, rd, rn, _ = Rr.generate_rays(idx, intr_c, c2w_c)
        out = orc.forward(ro, rd, rn.reshape(-1), idx[:, 0], tuple(j.double() for j in jit), anneal=anneal, training=True)
        gt_rgb = images_c[idx[:, 0], idx[:, 1], idx[:, 2]]
        gt_depth = depths_c[idx[:, 0], idx[:, 1], idx[:, 2]].reshape(-1)
        ld = orc.loss_dict(out, gt_rgb, gt_depth)
        orc.zero_grad()
        sum(ld.values()).backward()
        opt_fields.step()
        if updated:  # nerfacto evaluates the proposal networks under no_grad on the other steps
            opt_prop.step()
        orc_loss.append(float(ld["rgb_loss"].detach()))
        if step % 50 == 49:
            print(f"[psnr parity] step {step + 1}: rgb loss HIP {hip_loss[-1]:.4e}, oracle {orc_loss[-1]:.4e}", flush=True)
        if step + 1 in CHECKPOINTS:
            checkpoints.append(evaluate())
    torch.set_num_threads(saved_threads)
    torch.cuda.synchronize()
    assert int(eng.skip_flag.sum()) == 0

    # Two chaotic trajectories are compared (the HIP step is not bitwise reproducible either: float atomics), so a single
    # end point scatters by +-0.25 dB between runs of the SAME build; the mean over four checkpoints of the last 75 steps
    # is the statistic (measured over repeated runs: |delta| of a single checkpoint 0.01 ... 0.26 dB).
    ck = np.array(checkpoints)
    psnr_hip, psnr_orc, psnr_hip_t, psnr_orc_t = (float(v) for v in ck[:, :4].mean(axis=0))
    print(f"[psnr parity] checkpoints {sorted(CHECKPOINTS)}: held-out HIP/oracle {ck[:, 0].round(3).tolist()} / "
          f"{ck[:, 1].round(3).tolist()}, training views {ck[:, 2].round(3).tolist()} / {ck[:, 3].round(3).tolist()}")
    print(f"[psnr parity] training views (mean): HIP fp16 {psnr_hip_t:.3f} dB, oracle {psnr_orc_t:.3f} dB")
    tail_hip, tail_orc = float(np.mean(hip_loss[-50:])), float(np.mean(orc_loss[-50:]))
    head = float(np.mean(orc_loss[:10]))
    print(f"[psnr parity] held-out PSNR: HIP fp16 {psnr_hip:.3f} dB, oracle {psnr_orc:.3f} dB "
          f"(delta {psnr_hip - psnr_orc:+.3f}); rgb loss first 10 steps {head:.4e}, last 50 steps: HIP {tail_hip:.4e}, "
          f"oracle {tail_orc:.4e}; HIP vs oracle render: {ck[-1, 4]:.2f} dB")
    assert tail_orc < 0.5 * head, "the oracle run did not train"
    assert abs(tail_hip - tail_orc) <= 0.10 * tail_orc, (tail_hip, tail_orc)
    # per-step agreement while the trajectories are still close (before fp16 rounding has been amplified)
    early = np.abs(np.array(hip_loss[:20]) - np.array(orc_loss[:20])) / np.array(orc_loss[:20])
    assert early.max() < 0.05, early
    assert abs(psnr_hip - psnr_orc) <= 0.2, (psnr_hip, psnr_orc)
    assert abs(psnr_hip_t - psnr_orc_t) <= 0.2, (psnr_hip_t, psnr_orc_t)
