#!/usr/bin/env python3
"""What does the SE3 camera optimiser converge to when the poses it is given are EXACT?

The reference configures `CameraOptimizerConfig(mode='SE3')` with Adam(lr 1e-4 -> 1e-5, eps 1e-15)
(/root/reference/nerf_vo/mapping/nerfstudio.py:64,93-99) and exports multiply(correction, c2w) (:205-216).  At full
size (192 keyframes, 8192 iterations) the exported poses end ~3.5e-3 rad away from exact ones (BENCH_r04 render_psnr).
This study runs the HIP engine AND the float64 oracle (fp16 storage emulated) with the optimiser ON from identical
initial parameters on identical injected rays / jitters -- the scene of tests/helpers/psnr_long_parity.py, exact poses --
and records the learnt corrections every `--every` steps.  If both walk away from zero at the same rate, the floor is the
optimiser's own (Adam's normalised steps on a gradient that is noise around zero), not a defect of the kernels.

The report splits the corrections into their MEAN RIGID MOTION (a gauge: moving every camera by the same transform is
absorbed by the field) and the residual, and prints the reference scale lr x sqrt(steps) of a sign-random walk.

  --phase hip     (GPU box)  -> <dir>/init.pt, <dir>/hip.pt
  --phase oracle  (any CPU)  -> <dir>/oracle.pt      (resumable: --resume)
  --phase report
Lives under tests/ because it executes the oracle; run by hand, report committed under profiles/."""
import argparse
import math
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import psnr_long_parity as P  # noqa: E402  (scene, views, ray stream)


def _engine_lr_camera(cfg_lr, cfg_lr_final, max_iters, step):
    t = min(max(step / max(max_iters, 1), 0.0), 1.0)
    return math.exp(math.log(cfg_lr) * (1 - t) + math.log(cfg_lr_final) * t)


def phase_hip(a):
    import __graft_entry__ as entry

    entry.build()
    from nerf_vo_amd.engine import EngineConfig, GridConfig, NerfactoEngine
    from nerf_vo_amd.mapping.dataset import DynamicDataset, opencv_to_opengl
    from nerf_vo_amd.synthetic import make_sequence

    dev = torch.device("cuda:0")
    torch.manual_seed(11 + a.seed)
    seq = make_sequence(P.ORBIT, P.H, P.W, device=dev)
    pick = torch.tensor(P.TRAIN_FRAMES, device=dev)
    seq = {k: v[pick] for k, v in seq.items()}
    ds = DynamicDataset(num_frames=P.N_IMG, frame_height=P.H, frame_width=P.W, device=dev, use_normals=False)
    ds.update({"keyframe_indices": torch.arange(P.N_IMG), "camera_intrinsics": seq["camera_intrinsics"],
               "camera_extrinsics": opencv_to_opengl(seq["camera_extrinsics"]), "frames_color": seq["frames_color"],
               "frames_depth": seq["frames_depth"]})
    intr, c2w = ds.camera_intrinsics, ds.camera_extrinsics[:, :3, :4].contiguous()
    images, depths = ds.frames_color, ds.frames_depth
    cfg = EngineConfig(num_images=P.N_IMG, num_rays=a.rays, main_grid=GridConfig(*P.GRIDS["main"]),
                       proposal_grids=tuple(GridConfig(*g) for g in P.GRIDS["props"]), optimize_poses=True,
                       camera_mode="SE3", max_num_iterations=a.steps, seed=1337 + a.seed)
    eng = NerfactoEngine(cfg, dev)
    torch.save({"params": eng.params.detach().cpu(), "segments": dict(eng.segments), "intr": intr.cpu(), "c2w": c2w.cpu(),
                "images": images.cpu(), "depths": depths.cpu(), "density_bias": cfg.density_bias,
                "lr": (cfg.lr_fields, cfg.lr_proposal, cfg.lr_camera, cfg.lr_camera_final), "betas": cfg.adam_betas,
                "eps": cfg.adam_eps, "anneal": [eng.anneal_at(s) for s in range(a.steps)], "rays": a.rays, "steps": a.steps,
                "seed": a.seed, "penalties": (cfg.camera_trans_l2_penalty, cfg.camera_rot_l2_penalty)},
               os.path.join(a.dir, "init.pt"))
    out = {"pose": {}, "updated": [], "loss": []}
    for step, (idx, jit) in enumerate(P._ray_stream(a.steps, a.rays, a.seed)):
        upd = eng.train_step(idx.to(dev), intr, c2w, images, depths, jitters=tuple(j.to(dev) for j in jit))
        out["updated"].append(bool(upd))
        if (step + 1) % a.every == 0 or step + 1 == a.steps:
            out["pose"][step + 1] = eng.view("camera_opt.pose_adjustment").detach().view(P.N_IMG, 6).cpu().clone()
            out["loss"].append((step + 1, eng.loss_dict()["rgb_loss"]))
    torch.cuda.synchronize()
    out["skipped"] = int(eng.skip_flag.sum())
    torch.save(out, os.path.join(a.dir, "hip.pt"))
    print(f"[hip] {a.steps} steps, last rgb loss {out['loss'][-1][1]:.3e}, skipped flags {out['skipped']}", flush=True)


def phase_oracle(a):
    from oracle import mlp as omlp
    from oracle import rays as Rr
    from oracle.nerfacto import NerfactoOracle, OracleConfig

    torch.set_num_threads(a.threads)
    init = torch.load(os.path.join(a.dir, "init.pt"))
    steps, rays = init["steps"], init["rays"]
    orc = NerfactoOracle(OracleConfig(num_images=P.N_IMG, main_grid=P.GRIDS["main"], proposal_grids=P.GRIDS["props"],
                                      density_bias=init["density_bias"], emulate_fp16=True))
    p = init["params"].double()

    def seg(name):
        o, s, _ = init["segments"][name]
        return p[o:o + s]

    nb, npk = omlp.mlp_n_params(32, 16, 64, 1), omlp.mlp_n_params(10, 1, 16, 1)
    orc.params = {"base_mlp": seg("field.base")[:nb].clone(), "base_grid": seg("field.base")[nb:].clone().view(-1, 2),
                  "color_mlp": seg("field.color").clone(), "embedding": seg("field.embedding").clone().view(P.N_IMG, 32)}
    for k in range(2):
        orc.params[f"prop{k}_mlp"] = seg(f"proposal.{k}")[:npk].clone()
        orc.params[f"prop{k}_grid"] = seg(f"proposal.{k}")[npk:].clone().view(-1, 2)
    for v in orc.params.values():
        v.requires_grad_(True)
    pose = torch.zeros(P.N_IMG, 6, dtype=torch.float64, requires_grad=True)
    lr_f, lr_p, lr_c, lr_c_end = init["lr"]
    opt_fields = torch.optim.Adam([orc.params[k] for k in ("base_mlp", "base_grid", "color_mlp", "embedding")],
                                  lr=lr_f, betas=init["betas"], eps=init["eps"])
    opt_prop = torch.optim.Adam([orc.params[k] for k in ("prop0_mlp", "prop0_grid", "prop1_mlp", "prop1_grid")],
                                lr=lr_p, betas=init["betas"], eps=init["eps"])
    opt_cam = torch.optim.Adam([pose], lr=lr_c, betas=init["betas"], eps=init["eps"])
    hip = torch.load(os.path.join(a.dir, "hip.pt"))
    images, depths = init["images"].double(), init["depths"].double()
    intr, c2w = init["intr"].double(), init["c2w"].double()
    tp, rp = init["penalties"]
    state_path = os.path.join(a.dir, "oracle_state.pt")
    out = {"pose": {}, "loss": []}
    start = 0
    if a.resume and os.path.exists(state_path):
        st = torch.load(state_path)
        for k, v in st["params"].items():
            orc.params[k].data.copy_(v)
        pose.data.copy_(st["pose"])
        opt_fields.load_state_dict(st["opt_fields"])
        opt_prop.load_state_dict(st["opt_prop"])
        opt_cam.load_state_dict(st["opt_cam"])
        out, start = st["out"], st["step"]
        print(f"[oracle] resumed at step {start}", flush=True)
    t0 = time.time()
    for step, (idx, jit) in enumerate(P._ray_stream(steps, rays, init.get("seed", 0))):
        if step < start:
            continue
        ro, rd, rn, _ = Rr.generate_rays(idx, intr, c2w)
        corr = Rr.exp_map_se3(pose)[idx[:, 0]]
        ro2, rd2 = Rr.apply_pose_correction(ro, rd, corr)
        o = orc.forward(ro2, rd2, rn.reshape(-1), idx[:, 0], tuple(j.double() for j in jit), anneal=init["anneal"][step], training=True)
        ld = orc.loss_dict(o, images[idx[:, 0], idx[:, 1], idx[:, 2]], depths[idx[:, 0], idx[:, 1], idx[:, 2]].reshape(-1))
        ld["camera_opt_regularizer"] = Rr.camera_opt_regularizer(pose, tp, rp)
        orc.zero_grad()
        pose.grad = None
        sum(ld.values()).backward()
        opt_fields.step()
        if hip["updated"][step]:
            opt_prop.step()
        for gp in opt_cam.param_groups:  # ExponentialDecay lr 1e-4 -> 1e-5 over max_num_iterations, as the engine
            gp["lr"] = _engine_lr_camera(lr_c, lr_c_end, steps, step)
        opt_cam.step()
        done = step + 1
        if done % a.every == 0 or done == steps:
            out["pose"][done] = pose.detach().clone()
            out["loss"].append((done, float(ld["rgb_loss"].detach())))
            print(f"[oracle] step {done}: rgb loss {float(ld['rgb_loss']):.4e}, rot rms {float(pose[:, 3:].detach().pow(2).mean().sqrt()):.3e}, "
                  f"{time.time() - t0:.0f} s", flush=True)
            torch.save({"params": {k: v.detach() for k, v in orc.params.items()}, "pose": pose.detach(),
                        "opt_fields": opt_fields.state_dict(), "opt_prop": opt_prop.state_dict(),
                        "opt_cam": opt_cam.state_dict(), "out": out, "step": done}, state_path)
    torch.save(out, os.path.join(a.dir, "oracle.pt"))


def split_gauge(pose):
    """pose [F,6] tangents (translation | rotation), small: first-order split into the mean rigid motion and the residual.
    Returns (rotation of the mean, mean |rotation| raw, mean |rotation| of the residual, the same three for translation)."""
    pose = pose.double()
    rot, tr = pose[:, 3:], pose[:, :3]
    mr, mt = rot.mean(0), tr.mean(0)
    return (float(mr.norm()), float(rot.norm(dim=1).mean()), float((rot - mr).norm(dim=1).mean()),
            float(mt.norm()), float(tr.norm(dim=1).mean()), float((tr - mt).norm(dim=1).mean()))


def phase_report(a):
    init = torch.load(os.path.join(a.dir, "init.pt"))
    hip = torch.load(os.path.join(a.dir, "hip.pt"))
    orc = torch.load(os.path.join(a.dir, "oracle.pt")) if os.path.exists(os.path.join(a.dir, "oracle.pt")) else \
        torch.load(os.path.join(a.dir, "oracle_state.pt"))["out"]
    lr_c, lr_end = init["lr"][2], init["lr"][3]
    print(f"SE3 camera optimiser on EXACT poses: {init['steps']} steps x {init['rays']} rays, {P.N_IMG} views {P.W}x{P.H}, HIP engine vs "
          f"float64 oracle (fp16 storage emulated), identical initial parameters / rays / jitters; Adam lr {lr_c:g} -> {lr_end:g}")
    print("step | rotation of the corrections [rad]: mean |w| raw / residual after removing the mean rigid motion / the mean itself "
          "| same for translation | HIP then oracle | lr-sum scale (sum of lr_t over the steps: the distance Adam's unit steps cover "
          "if every step had the same sign) and sqrt-scale (sign-random walk)")
    lrs = np.array([_engine_lr_camera(lr_c, lr_end, init["steps"], s) for s in range(init["steps"])])
    rows = []
    for s in sorted(set(hip["pose"]) & set(orc["pose"])):
        h, o = split_gauge(hip["pose"][s]), split_gauge(orc["pose"][s])
        lin, rw = float(lrs[:s].sum()), float(np.sqrt((lrs[:s] ** 2).sum()))
        rows.append((s, *h, *o, lin, rw))
        print(f"{s:5d} | HIP rot {h[1]:.2e} / {h[2]:.2e} / {h[0]:.2e}  tr {h[4]:.2e} / {h[5]:.2e} / {h[3]:.2e} | oracle rot "
              f"{o[1]:.2e} / {o[2]:.2e} / {o[0]:.2e}  tr {o[4]:.2e} / {o[5]:.2e} / {o[3]:.2e} | {lin:.2e}  {rw:.2e}")
    last = rows[-1]
    print(f"end: HIP rotation residual {last[3]:.3e} rad, oracle {last[9]:.3e} rad (ratio {last[3] / max(last[9], 1e-30):.2f}); "
          f"sign-random-walk scale {last[-1]:.3e} rad")
    import json
    print(json.dumps({"rows": rows}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--phase", choices=("hip", "oracle", "report"), required=True)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--rays", type=int, default=512)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--resume", action="store_true")
    args = ap.parse_args()
    if args.dir is None:
        args.dir = os.path.join(ROOT, "gpurun_out", f"se3_floor_s{args.seed}")
    os.makedirs(args.dir, exist_ok=True)
    {"hip": phase_hip, "oracle": phase_oracle, "report": phase_report}[args.phase](args)
