#!/usr/bin/env python3
"""Long like-for-like PSNR run: the HIP engine (fp16 operands, fp32 accumulate) against the CPU oracle that emulates
tiny-cuda-nn's 16-bit storage points, trained from IDENTICAL initial parameters on IDENTICAL injected rays and jitters
for >= 2000 steps at 1024 rays, both rendering the same held-out and training views at checkpoints.  The long form of
tests/test_psnr_parity_gpu.py (300 steps, 128 rays); lives under tests/ because it executes the oracle.  Not collected
by pytest: it is run by hand and its report is committed under profiles/.

Three phases, because the oracle needs CPU-hours and the GPU box grants 20-minute leases:

  --phase hip     (GPU box)  builds the engine, writes its initial parameters + the keyframes to <dir>/init.pt, trains,
                             renders at the checkpoints -> <dir>/hip.pt
  --phase oracle  (any CPU)  loads init.pt, trains the float64 oracle on the same ray / jitter stream (torch CPU generator,
                             seed fixed), renders the same views -> <dir>/oracle.pt   (resumable: --resume)
  --phase report             reads both, prints the table
  --phase summary            over --seeds: mean +- spread of both sides' end points and the paired differences

Sizes: main grid 16 levels x 2^14, proposal grids 5 levels x 2^12 (the oracle has to finish), production kernels,
MLP shapes, samplers, losses and Adam."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

# 12 training views, 15 degrees apart along the orbit, and 4 held-out views half way BETWEEN training views (the 64x48
# frames see 90 degrees, so every held-out pixel is covered by several training views): a 96-frame orbit, training frames
# 0, 4, ..., 44, held-out frames 6, 18, 30, 42.  Buffer order: training views first, then the held-out ones.
N_IMG, N_HELD, H, W = 12, 4, 48, 64
ORBIT, TRAIN_FRAMES, HELD_FRAMES = 96, list(range(0, 48, 4)), [6, 18, 30, 42]
GRIDS = dict(main=(16, 14, 16, 512), props=((5, 12, 16, 64), (5, 12, 16, 128)))


def _psnr(pred, gt):
    return float(-10.0 * torch.log10(torch.mean((pred.double().cpu() - gt.double().cpu()) ** 2)))


def _views(view_ids, stride):
    cams, ys, xs = torch.meshgrid(torch.tensor(view_ids), torch.arange(0, H, stride), torch.arange(0, W, stride), indexing="ij")
    return torch.stack([cams, ys, xs], dim=-1).reshape(-1, 3)


def _ray_stream(steps, rays, seed=0):
    """The injected (pixel, jitter) stream both phases consume: ONE CPU generator, same draws in the same order."""
    g = torch.Generator().manual_seed(5 + 7919 * int(seed))
    for _ in range(steps):
        idx = torch.stack([torch.randint(0, N_IMG, (rays,), generator=g), torch.randint(0, H, (rays,), generator=g),
                           torch.randint(0, W, (rays,), generator=g)], dim=1)
        yield idx, tuple(torch.rand(rays, generator=g) for _ in range(3))


def phase_hip(a):
    import __graft_entry__ as entry

    entry.build()
    from nerf_vo_amd.engine import EngineConfig, GridConfig, NerfactoEngine
    from nerf_vo_amd.mapping.dataset import DynamicDataset, opencv_to_opengl
    from nerf_vo_amd.synthetic import make_sequence

    dev = torch.device("cuda:0")
    torch.manual_seed(11 + a.seed)
    seq = make_sequence(ORBIT, H, W, device=dev)
    pick = torch.tensor(TRAIN_FRAMES + HELD_FRAMES, device=dev)  # (the last views of the buffer are held out)
    seq = {k: v[pick] for k, v in seq.items()}
    ds = DynamicDataset(num_frames=N_IMG + N_HELD, frame_height=H, frame_width=W, device=dev, use_normals=False)
    ds.update({"keyframe_indices": torch.arange(N_IMG + N_HELD), "camera_intrinsics": seq["camera_intrinsics"],
               "camera_extrinsics": opencv_to_opengl(seq["camera_extrinsics"]), "frames_color": seq["frames_color"],
               "frames_depth": seq["frames_depth"]})
    intr, c2w = ds.camera_intrinsics, ds.camera_extrinsics[:, :3, :4].contiguous()
    images, depths = ds.frames_color, ds.frames_depth
    eng = NerfactoEngine(EngineConfig(num_images=N_IMG, num_rays=a.rays, main_grid=GridConfig(*GRIDS["main"]),
                                      proposal_grids=tuple(GridConfig(*g) for g in GRIDS["props"]),
                                      dynamic_loss_scale=not a.static_loss_scale, seed=1337 + a.seed), dev)
    torch.save({"params": eng.params.detach().cpu(), "segments": dict(eng.segments), "intr": intr.cpu(), "c2w": c2w.cpu(),
                "images": images.cpu(), "depths": depths.cpu(), "density_bias": eng.cfg.density_bias,
                "lr": (eng.cfg.lr_fields, eng.cfg.lr_proposal), "betas": eng.cfg.adam_betas, "eps": eng.cfg.adam_eps,
                "anneal": [eng.anneal_at(s) for s in range(a.steps)], "rays": a.rays, "steps": a.steps, "seed": a.seed},
               os.path.join(a.dir, "init.pt"))
    from oracle import rays as Rr  # (the view rays of both phases come from the same CPU ray generator)

    held, train = _views(list(range(N_IMG, N_IMG + N_HELD)), 1), _views(list(range(N_IMG)), 2)

    def render(idx):
        ro, rd, rn, _ = Rr.generate_rays(idx, intr.cpu().double(), c2w.cpu().double())
        rn = rn.reshape(-1)
        return torch.cat([eng.render_rays(ro[c:c + 2048].float().to(dev), rd[c:c + 2048].float().to(dev),
                                          rn[c:c + 2048].float().to(dev))["rgb"].cpu() for c in range(0, idx.shape[0], 2048)])

    out = {"checkpoints": {}, "loss": [], "updated": []}
    t0 = time.time()
    for step, (idx, jit) in enumerate(_ray_stream(a.steps, a.rays, a.seed)):
        upd = eng.train_step(idx.to(dev), intr, c2w, images, depths, jitters=tuple(j.to(dev) for j in jit))
        out["updated"].append(bool(upd))
        if step % 25 == 24 or step + 1 == a.steps:
            out["loss"].append((step + 1, eng.loss_dict()["rgb_loss"]))
        if (step + 1) % a.every == 0 or step + 1 == a.steps:
            out["checkpoints"][step + 1] = {"held": render(held), "train": render(train)}
            print(f"[hip] step {step + 1}: rgb loss {eng.loss_dict()['rgb_loss']:.4e}, {time.time() - t0:.0f} s", flush=True)
    torch.cuda.synchronize()
    out["skipped"] = int(eng.skip_flag.sum())
    out["loss_scale_end"] = eng.current_loss_scale()
    torch.save(out, os.path.join(a.dir, "hip.pt"))


def phase_oracle(a):
    from oracle import mlp as omlp
    from oracle import rays as Rr
    from oracle.nerfacto import NerfactoOracle, OracleConfig

    torch.set_num_threads(a.threads)
    init = torch.load(os.path.join(a.dir, "init.pt"))
    steps, rays = init["steps"], init["rays"]
    orc = NerfactoOracle(OracleConfig(num_images=N_IMG, main_grid=GRIDS["main"], proposal_grids=GRIDS["props"],
                                      density_bias=init["density_bias"], emulate_fp16=True))
    p = init["params"].double()

    def seg(name):
        o, s, _ = init["segments"][name]
        return p[o:o + s]

    nb, npk = omlp.mlp_n_params(32, 16, 64, 1), omlp.mlp_n_params(10, 1, 16, 1)
    orc.params = {"base_mlp": seg("field.base")[:nb].clone(), "base_grid": seg("field.base")[nb:].clone().view(-1, 2),
                  "color_mlp": seg("field.color").clone(), "embedding": seg("field.embedding").clone().view(N_IMG, 32)}
    for k in range(2):
        orc.params[f"prop{k}_mlp"] = seg(f"proposal.{k}")[:npk].clone()
        orc.params[f"prop{k}_grid"] = seg(f"proposal.{k}")[npk:].clone().view(-1, 2)
    for v in orc.params.values():
        v.requires_grad_(True)
    opt_fields = torch.optim.Adam([orc.params[k] for k in ("base_mlp", "base_grid", "color_mlp", "embedding")],
                                  lr=init["lr"][0], betas=init["betas"], eps=init["eps"])
    opt_prop = torch.optim.Adam([orc.params[k] for k in ("prop0_mlp", "prop0_grid", "prop1_mlp", "prop1_grid")],
                                lr=init["lr"][1], betas=init["betas"], eps=init["eps"])
    hip = torch.load(os.path.join(a.dir, "hip.pt"))  # (which steps refreshed the proposal networks: the engine's schedule)
    images, depths = init["images"].double(), init["depths"].double()
    intr, c2w = init["intr"].double(), init["c2w"].double()
    held, train = _views(list(range(N_IMG, N_IMG + N_HELD)), 1), _views(list(range(N_IMG)), 2)

    def render(idx):
        ro, rd, rn, _ = Rr.generate_rays(idx, intr, c2w)
        with torch.no_grad():
            return orc.forward(ro, rd, rn.reshape(-1), idx[:, 0].clamp(max=N_IMG - 1), None, anneal=1.0, training=False)["rgb"].float()

    state_path = os.path.join(a.dir, "oracle_state.pt")
    out = {"checkpoints": {}, "loss": []}
    start = 0
    if a.resume and os.path.exists(state_path):
        st = torch.load(state_path)
        for k, v in st["params"].items():
            orc.params[k].data.copy_(v)
        opt_fields.load_state_dict(st["opt_fields"])
        opt_prop.load_state_dict(st["opt_prop"])
        out, start = st["out"], st["step"]
        print(f"[oracle] resumed at step {start}", flush=True)
    t0 = time.time()
    for step, (idx, jit) in enumerate(_ray_stream(steps, rays, init.get("seed", 0))):
        if step < start:
            continue
        ro, rd, rn, _ = Rr.generate_rays(idx, intr, c2w)
        o = orc.forward(ro, rd, rn.reshape(-1), idx[:, 0], tuple(j.double() for j in jit), anneal=init["anneal"][step], training=True)
        ld = orc.loss_dict(o, images[idx[:, 0], idx[:, 1], idx[:, 2]], depths[idx[:, 0], idx[:, 1], idx[:, 2]].reshape(-1))
        orc.zero_grad()
        sum(ld.values()).backward()
        opt_fields.step()
        if hip["updated"][step]:  # nerfacto evaluates the proposal networks under no_grad on the other steps
            opt_prop.step()
        if step % 25 == 24 or step + 1 == steps:
            out["loss"].append((step + 1, float(ld["rgb_loss"].detach())))
        done = step + 1
        if done % a.every == 0 or done == steps:
            out["checkpoints"][done] = {"held": render(held), "train": render(train)}
            print(f"[oracle] step {done}: rgb loss {float(ld['rgb_loss']):.4e}, {time.time() - t0:.0f} s", flush=True)
            torch.save({"params": {k: v.detach() for k, v in orc.params.items()}, "opt_fields": opt_fields.state_dict(),
                        "opt_prop": opt_prop.state_dict(), "out": out, "step": done}, state_path)
    torch.save(out, os.path.join(a.dir, "oracle.pt"))


def phase_report(a):
    init = torch.load(os.path.join(a.dir, "init.pt"))
    hip, orc = torch.load(os.path.join(a.dir, "hip.pt")), torch.load(os.path.join(a.dir, "oracle.pt"))
    images = init["images"]
    held, train = _views(list(range(N_IMG, N_IMG + N_HELD)), 1), _views(list(range(N_IMG)), 2)
    gt_h, gt_t = images[held[:, 0], held[:, 1], held[:, 2]], images[train[:, 0], train[:, 1], train[:, 2]]
    print(f"HIP engine vs the fp16-emulating float64 oracle: {init['steps']} steps x {init['rays']} rays, identical initial parameters, "
          f"rays and jitters; {N_IMG} training views {W}x{H}, {N_HELD} held-out views; HIP skipped-step flags at the end: "
          f"{hip['skipped']}, loss scale {hip['loss_scale_end']}")
    print("step | held-out PSNR HIP / oracle (delta) | training views HIP / oracle (delta) | HIP-vs-oracle render (held-out)")
    rows = []
    for s in sorted(set(hip["checkpoints"]) & set(orc["checkpoints"])):
        h, o = hip["checkpoints"][s], orc["checkpoints"][s]
        r = (s, _psnr(h["held"], gt_h), _psnr(o["held"], gt_h), _psnr(h["train"], gt_t), _psnr(o["train"], gt_t), _psnr(h["held"], o["held"]))
        rows.append(r)
        print(f"{r[0]:5d} | {r[1]:6.2f} / {r[2]:6.2f} ({r[1] - r[2]:+.2f}) | {r[3]:6.2f} / {r[4]:6.2f} ({r[3] - r[4]:+.2f}) | {r[5]:6.2f}")
    tail = np.array(rows[-4:])
    print(f"mean of the last {len(tail)} checkpoints: held-out HIP {tail[:, 1].mean():.3f} dB, oracle {tail[:, 2].mean():.3f} dB "
          f"(delta {tail[:, 1].mean() - tail[:, 2].mean():+.3f}); training views HIP {tail[:, 3].mean():.3f} dB, oracle "
          f"{tail[:, 4].mean():.3f} dB (delta {tail[:, 3].mean() - tail[:, 4].mean():+.3f})")
    lh, lo = dict(hip["loss"]), dict(orc["loss"])
    common = sorted(set(lh) & set(lo))[-20:]
    print(f"rgb loss, mean over the last {len(common)} logged steps: HIP {np.mean([lh[s] for s in common]):.4e}, "
          f"oracle {np.mean([lo[s] for s in common]):.4e}")
    print(json.dumps({"rows": rows}))


def _end_points(d, last=4):
    """mean PSNR of the last checkpoints of one seed's pair of runs: (held HIP, held oracle, train HIP, train oracle)"""
    init = torch.load(os.path.join(d, "init.pt"))
    hip, orc = torch.load(os.path.join(d, "hip.pt")), torch.load(os.path.join(d, "oracle.pt"))
    images = init["images"]
    held, train = _views(list(range(N_IMG, N_IMG + N_HELD)), 1), _views(list(range(N_IMG)), 2)
    gt_h, gt_t = images[held[:, 0], held[:, 1], held[:, 2]], images[train[:, 0], train[:, 1], train[:, 2]]
    steps = sorted(set(hip["checkpoints"]) & set(orc["checkpoints"]))[-last:]
    rows = np.array([[_psnr(hip["checkpoints"][s]["held"], gt_h), _psnr(orc["checkpoints"][s]["held"], gt_h),
                      _psnr(hip["checkpoints"][s]["train"], gt_t), _psnr(orc["checkpoints"][s]["train"], gt_t)] for s in steps])
    return rows.mean(axis=0), steps


def phase_summary(a):
    """Over the seeds: mean +- sample standard deviation of each side's end point, and the paired differences."""
    base = a.dir.rsplit("_s", 1)[0]
    pts = []
    for sd in a.seeds:
        e, steps = _end_points(f"{base}_s{sd}")
        pts.append(e)
        print(f"seed {sd}: held-out HIP {e[0]:.3f} / oracle {e[1]:.3f} dB (delta {e[0] - e[1]:+.3f}); training views HIP {e[2]:.3f} / "
              f"oracle {e[3]:.3f} dB (delta {e[2] - e[3]:+.3f})   [mean of checkpoints {steps}]")
    pts = np.array(pts)
    m, sdv = pts.mean(axis=0), pts.std(axis=0, ddof=1) if len(pts) > 1 else np.zeros(4)
    dh, dt = pts[:, 0] - pts[:, 1], pts[:, 2] - pts[:, 3]
    print(f"held-out:        HIP {m[0]:.3f} +- {sdv[0]:.3f} dB, oracle {m[1]:.3f} +- {sdv[1]:.3f} dB over {len(pts)} seeds; "
          f"paired delta {dh.mean():+.3f} +- {dh.std(ddof=1) if len(dh) > 1 else 0.0:.3f} dB")
    print(f"training views:  HIP {m[2]:.3f} +- {sdv[2]:.3f} dB, oracle {m[3]:.3f} +- {sdv[3]:.3f} dB; "
          f"paired delta {dt.mean():+.3f} +- {dt.std(ddof=1) if len(dt) > 1 else 0.0:.3f} dB")
    print(json.dumps({"seeds": list(a.seeds), "end_points": pts.tolist(), "mean": m.tolist(), "std": sdv.tolist(),
                      "held_delta": dh.tolist(), "train_delta": dt.tolist()}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--phase", choices=("hip", "oracle", "report", "summary"), required=True)
    ap.add_argument("--seed", type=int, default=0, help="initial parameters (1337 + seed) and ray / jitter stream")
    ap.add_argument("--seeds", type=int, nargs="+", default=[0, 1, 2], help="(summary) the seeds to combine")
    ap.add_argument("--dir", default=None)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--every", type=int, default=250)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--static-loss-scale", action="store_true")
    args = ap.parse_args()
    if args.dir is None:
        args.dir = os.path.join(ROOT, "gpurun_out", f"psnr_long_s{args.seed}")
    os.makedirs(args.dir, exist_ok=True)
    {"hip": phase_hip, "oracle": phase_oracle, "report": phase_report, "summary": phase_summary}[args.phase](args)
