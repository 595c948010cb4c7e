#!/usr/bin/env python3
"""Step time of the occupancy-grid ("instant-ngp") trainer through the pyngp facade on the synthetic room, on the
configuration the reference runs this back-end on (/root/reference/configs/nerf_slam_replica.yaml:6-19: 192 keyframes at
360x640, DROID-SLAM tracking with compute_covariances: True -- every keyframe arrives with a per-pixel depth VARIANCE,
/root/reference/nerf_vo/mapping/instant_ngp.py:77-100): `--cov varying` (the default) hands a smooth, non-constant variance
to update_training_images, so the variance gather and the Mahalanobis depth term run; `--cov ones` is the reference's
fallback without DROID-SLAM (plain L2 depth term, gather skipped).
python tools/ngp_bench.py [--steps 300] [--extrinsics 0|1] [--keyframes 192 --height 360 --width 640] [--cov varying|ones]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

entry.build()
from nerf_vo_amd import pyngp  # noqa: E402
from nerf_vo_amd.mapping.dataset import opencv_to_opengl  # noqa: E402
from nerf_vo_amd.synthetic import make_sequence  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=300)  # (past the 256 steps of full density-grid sweeps)
    ap.add_argument("--keyframes", type=int, default=192)
    ap.add_argument("--height", type=int, default=360)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--cov", choices=("varying", "ones"), default="varying",
                    help="per-pixel depth variance handed to update_training_images (varying: (0.02 + 0.03 depth)^2 with a smooth "
                         "image-space modulation -- the gather and the covariance-weighted depth term run; ones: plain L2)")
    ap.add_argument("--extrinsics", type=int, default=1)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--render-capacity", type=int, default=0, help="packed sample slots of an inference launch (0: NgpConfig)")
    ap.add_argument("--profile-render", action="store_true", help="per-kernel device time of the rendered frames")
    ap.add_argument("--render-frames", type=int, default=0, help="1200x680 views rendered through Testbed.render after training")
    ap.add_argument("--train-min-t", type=float, default=None, help="NgpConfig.train_min_transmittance (default 1e-4; 0 = every sample trains)")
    ap.add_argument("--first-round", type=int, default=0, help="samples per ray of the first inference round (0: NgpConfig.render_first_round)")
    a = ap.parse_args()
    run(a)


def run(a, quiet: bool = False):
    """a: namespace with steps, warmup, keyframes, extrinsics, profile.  Returns the JSON object of the --profile form
    (bench.py embeds it as its "ngp" section)."""
    say = (lambda *x: None) if quiet else print
    dev = torch.device("cuda:0")
    H, W = int(getattr(a, "height", 360)), int(getattr(a, "width", 640))
    cov_mode = getattr(a, "cov", "varying")
    seq = make_sequence(a.keyframes, H, W, device=dev, scene_scale=0.2)
    poses = seq["camera_extrinsics"].clone()
    poses[:, :3, 3] += 0.5
    tb = pyngp.Testbed(pyngp.TestbedMode.Nerf, 0)
    tb.create_empty_nerf_dataset(n_images=a.keyframes, nerf_scale=1.0, nerf_offset=np.zeros(3), aabb_scale=4)
    tb.reload_network_from_file("")
    tb.shall_train = True
    tb.nerf.training.optimize_extrinsics = bool(a.extrinsics)
    color = seq["frames_color"].permute(0, 2, 3, 1)
    color = torch.cat([color, torch.ones_like(color[..., :1])], dim=3)
    depth = seq["frames_depth"].permute(0, 2, 3, 1)
    if cov_mode == "ones":
        depths_cov = torch.ones_like(depth)
    else:
        # what a tracker's depth covariance looks like in shape: grows with depth, varies smoothly over the image, positive
        ys = torch.linspace(0.0, 3.14159, H, device=dev).view(1, H, 1, 1)
        xs = torch.linspace(0.0, 6.28318, W, device=dev).view(1, 1, W, 1)
        sigma = (0.02 + 0.03 * depth) * (1.0 + 0.25 * torch.sin(xs) * torch.cos(ys))
        depths_cov = (sigma * sigma).contiguous()
    tb.nerf.training.update_training_images(
        frame_ids=list(range(a.keyframes)), poses=opencv_to_opengl(poses)[:, :3], images=color.contiguous(),
        depths=depth.contiguous(), depths_cov=depths_cov, resolution=np.array([W, H]),
        principal_point=seq["camera_intrinsics"][0, 2:].cpu().numpy(), focal_length=seq["camera_intrinsics"][0, :2].cpu().numpy())
    if getattr(a, "train_min_t", None) is not None:
        tb.frame()
        tb._engine.cfg.train_min_transmittance = float(a.train_min_t)
    for _ in range(a.warmup):
        tb.frame()
    torch.cuda.synchronize()
    eng = tb._engine
    cap0 = (eng.graph_captures, eng.graph_capture_seconds)
    batches = set()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        batches.add(eng.rays_per_batch)
        tb.frame()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    window = {"first_step": eng.step - a.steps, "steps": a.steps, "ray_batches": sorted(batches),
              "graph_captures": eng.graph_captures - cap0[0],
              "graph_capture_ms": round((eng.graph_capture_seconds - cap0[1]) * 1e3, 2),
              "density_refresh": "scattered (cells / 4 per cascade uniform + as many occupied)"
              if eng.step - a.steps >= eng.cfg.density_warmup_steps else "every cell (first 256 steps) for part of the window"}
    say(f"timed window: {window}")
    if a.profile:
        import ctypes as C

        from nerf_vo_amd import _lib
        lib = _lib.lib()
        lib.nvo_profile_enable(1)
        graphed = eng.cfg.graph_step
        eng.cfg.graph_step = False  # (the per-kernel clocks bracket eager launches)
        for _ in range(32):
            tb.frame()
        torch.cuda.synchronize()
        eng.cfg.graph_step = graphed
        need = lib.nvo_profile_summary(None, 0)
        buf = C.create_string_buffer(int(need) + 16)
        lib.nvo_profile_summary(buf, len(buf))
        lib.nvo_profile_enable(0)
        rows = []
        for line in buf.value.decode().strip().splitlines():
            name, cnt, total = line.rsplit(",", 2)
            rows.append((name, int(cnt), float(total)))
        rows.sort(key=lambda r: -r[2])
        tot = sum(r[2] for r in rows)
        say(f"kernel time {tot / 32:.3f} ms/step over 32 steps")
        for name, cnt, total in rows[:24]:
            say(f"  {name:30s} launches {cnt:4d} avg {total / cnt * 1e3:9.1f} us  {100 * total / tot:5.1f} %")
    # inference through the facade, as evaluation/nerf_renderer.py drives it (colour + depth of one 1200x680 view)
    render = None
    if getattr(a, "render_frames", 0) > 0:
        import math

        if getattr(a, "first_round", 0):
            eng.cfg.render_first_round = int(a.first_round)
        if getattr(a, "render_capacity", 0):
            eng.cfg.render_capacity = int(a.render_capacity)
        fx = float(seq["camera_intrinsics"][0, 0]) * 1200.0 / W
        tb.fov_axis, tb.fov, tb.exposure = 0, 2.0 * math.degrees(math.atan(0.5 * 1200.0 / fx)), 0.0
        times = []
        for f in range(a.render_frames + 1):
            m = poses[f % a.keyframes].detach().cpu().numpy().astype(np.float64).copy()
            m[0:3, 1:3] *= -1
            tb.set_nerf_camera_matrix(m[[2, 0, 1]])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            tb.render_mode = pyngp.Shade
            shade = tb.render(width=1200, height=680, spp=1, linear=True)
            tb.render_mode = pyngp.Depth
            depth = tb.render(width=1200, height=680, spp=1, linear=True)
            times.append(time.perf_counter() - t1)
        ms = 1e3 * float(np.mean(times[1:]))
        if getattr(a, "profile_render", False):
            # per-kernel device time of two more frames (HIP events around every launch: the frame itself gets slower)
            import ctypes as C

            from nerf_vo_amd import _lib
            lib = _lib.lib()
            lib.nvo_profile_enable(1)
            for f in range(2):
                m = poses[(a.render_frames + 1 + f) % a.keyframes].detach().cpu().numpy().astype(np.float64).copy()
                m[0:3, 1:3] *= -1
                tb.set_nerf_camera_matrix(m[[2, 0, 1]])  # (a new view: the facade keeps the last pass)
                tb.render_mode = pyngp.Shade
                tb.render(width=1200, height=680, spp=1, linear=True)
                tb.render_mode = pyngp.Depth
                tb.render(width=1200, height=680, spp=1, linear=True)
            torch.cuda.synchronize()
            need = lib.nvo_profile_summary(None, 0)
            buf = C.create_string_buffer(int(need) + 16)
            lib.nvo_profile_summary(buf, len(buf))
            lib.nvo_profile_enable(0)
            rows = []
            for line in buf.value.decode().strip().splitlines():
                name, cnt, total = line.rsplit(",", 2)
                rows.append((name, int(cnt), float(total)))
            rows.sort(key=lambda r: -r[2])
            tot = sum(r[2] for r in rows)
            say(f"render kernel time {tot / 2:.2f} ms per frame (colour + depth) against {ms:.1f} ms wall")
            for name, cnt, total in rows[:16]:
                say(f"  {name:30s} launches/frame {cnt / 2:6.1f} avg {total / cnt * 1e3:9.1f} us  {100 * total / max(tot, 1e-12):5.1f} %")
        # quality of what was trained so far: views halfway between training cameras (the same trajectory sampled twice
        # as densely: odd frames) and training views, at the training resolution, float-MSE PSNR of the colour image
        def view_psnr(pose_cv, gt_chw, keyframe=None):
            if keyframe is not None:  # a training view: at the model's own (optimised) pose, as the reference renders keyframes
                tb.set_nerf_camera_matrix(tb.nerf.training.get_camera_extrinsics(keyframe))
            else:
                mm = pose_cv.detach().cpu().numpy().astype(np.float64).copy()
                mm[0:3, 1:3] *= -1
                tb.set_nerf_camera_matrix(mm[[2, 0, 1]])
            tb.fov = 2.0 * math.degrees(math.atan(0.5 * W / float(seq["camera_intrinsics"][0, 0])))
            tb.render_mode = pyngp.Shade
            img = np.clip(tb.render(width=W, height=H, spp=1, linear=True)[..., :3], 0.0, 1.0)
            mse = float(np.mean((img - gt_chw.permute(1, 2, 0).cpu().numpy()) ** 2))
            return 10.0 * math.log10(1.0 / max(mse, 1e-12))

        dense = make_sequence(2 * a.keyframes, H, W, device=dev, scene_scale=0.2)
        dposes = dense["camera_extrinsics"].clone()
        dposes[:, :3, 3] += 0.5
        same_path = bool(torch.allclose(dposes[::2], poses, atol=1e-5))
        held = [view_psnr(dposes[i], dense["frames_color"][i]) for i in range(1, 2 * a.keyframes, 2 * a.keyframes // 4)] if same_path else []
        seen = [view_psnr(poses[i], seq["frames_color"][i], keyframe=i) for i in range(0, a.keyframes, a.keyframes // 4)]
        render = {"resolution": [1200, 680], "frames": a.render_frames, "ms_per_frame_colour_and_depth": round(ms, 2),
                  "rays_per_sec": round(1200 * 680 / (ms * 1e-3)), "coverage": round(float((shade[..., 3] > 0.5).mean()), 4),
                  "psnr_after_steps": int(eng.step), "psnr_training_views_db": round(float(np.mean(seen)), 2),
                  "psnr_heldout_views_db": round(float(np.mean(held)), 2) if held else None,
                  "launch": f"eager, bundles sized to ~85 % of {int(eng.cfg.render_capacity or eng.cfg.capacity)} packed samples, both modes from one pass (host copies of both images included)"}
        say(f"render 1200x680 (colour + depth): {ms:.1f} ms per frame, {render['rays_per_sec'] / 1e6:.1f} M rays/s, "
            f"coverage {render['coverage']}, depth median {float(np.median(depth[..., 0])):.3f}; after {eng.step} steps PSNR "
            f"{render['psnr_training_views_db']} dB on training views, {render['psnr_heldout_views_db']} dB on views between them")
    n = eng.samples_last_step()
    say(f"extrinsics={a.extrinsics}: {dt * 1e3:.3f} ms/step, {n} packed samples in the last step "
        f"({n / dt / 1e6:.1f} M samples/s), losses {eng.loss_dict()}")
    if a.profile:
        # one JSON line in the shape of bench.py's (occupancy-grid back-end; SURVEY.md section 8d bytes: 588 B per
        # packed sample for the gather, 1100 B for the scatter)
        import json

        per = {name: total / cnt * 1e-3 for name, cnt, total in rows}  # seconds per launch
        cap = eng.cfg.capacity

        def pmc_traffic(name):
            """HBM bytes per launch from the committed PMC summary of this command (tools/collect_profile_evidence.sh:
            separate --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE corrected as tools/pmc_traffic.py describes)."""
            import glob
            import os

            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            for path in sorted(glob.glob(os.path.join(root, "profiles", "*_pmc_ngp_fetch_write_per_kernel.json")), reverse=True):
                for k in json.load(open(path)).get("kernels", []):
                    if k.get("bench_name") == name:
                        fetch = k.get("FETCH_SIZE_KB_corrected", k["FETCH_SIZE_KB_per_launch"])
                        return int((fetch + k["WRITE_SIZE_KB_per_launch"]) * 1024), os.path.basename(path)
            return None, None

        roof = None
        plan = eng._fused_adam_plan()
        fused_n = (plan[1] - plan[0]) if plan else 0
        # the record pass steps the parameters of its streamed hashed levels itself: Adam at 28 B per parameter (bench.py's
        # figure) + the weight average folded in (fp32 average read + written, 16-bit copy: 10 B)
        per_param = 28 + (10 if eng.cfg.ema_decay > 0.0 else 0)
        for name, bytes_per_sample in (("grid_bwd_stream[L16]", 1100), ("grid_fwd[L16]", 588)):
            if name in per:
                scatter = cap * bytes_per_sample
                b = scatter + (per_param * fused_n if name == "grid_bwd_stream[L16]" else 0)
                traffic, src = pmc_traffic(name)
                roof = roof or {"kernel": name, "bound": "hbm", "achieved": round(b / per[name] / 1e9, 1), "peak": 8000.0,
                                "unit": "GB/s", "frac": round(b / per[name] / 1e9 / 8000.0, 4), "traffic": traffic,
                                "traffic_source": src,
                                "avg_launch_us": round(per[name] * 1e6, 1), "algorithmic_bytes_per_launch": b,
                                "bytes_model": f"SURVEY.md 8d: {bytes_per_sample} B x {cap} packed samples" + (
                                    f" + {per_param} B x {fused_n} parameters stepped and averaged inside the pass"
                                    if (fused_n and name == "grid_bwd_stream[L16]") else ""),
                                "samples_only": {"algorithmic_bytes_per_launch": scatter,
                                                 "frac": round(scatter / per[name] / 1e9 / 8000.0, 4)}}
        out = {"metric": "packed training samples/sec (occupancy-grid back-end)", "value": n / dt,
                          "unit": "samples/s", "n_gpus": 1, "ms_per_step": dt * 1e3, "rays_per_batch": eng.rays_per_batch,
                          "dtype": "f16", "data": "synthetic",
                          "config": {"workload": f"pyngp.Testbed.frame(): {a.keyframes} keyframes {W}x{H} (configs/nerf_slam_replica.yaml:14-19), "
                                                 f"per-pixel depth variance {'non-constant: the covariance-weighted depth term and its gather run' if cov_mode != 'ones' else 'all ones (plain L2)'}, aabb_scale 4, "
                                                 f"capacity {cap} packed samples, extrinsics optimisation "
                                                 f"{'on' if a.extrinsics else 'off'} (camera step every {eng.cfg.extrinsic_update_every} training steps), weight EMA, "
                                                 f"adaptive ray batch, random background {'on' if eng.cfg.random_background else 'off'}, "
                                                 f"untrained cells marked",
                                     "launch": "hipGraph replay: ONE graph per step (per ray count), density-grid refresh "
                                               "eager every 16th step" if eng.cfg.graph_step else "eager"},
                          "window": window, "render": render, "march_us": round(per.get("occ_march", 0.0) * 1e6, 1), "roofline": roof,
                          "kernel_ms_per_step": round(tot / 32, 4),
                          "kernel_table": [{"kernel": nm, "launches_per_step": round(c / 32, 2), "avg_launch_us": round(t / c * 1e3, 1)}
                                           for nm, c, t in rows[:12]]}
        say(json.dumps(out))
        return out
    return {"ms_per_step": dt * 1e3, "value": n / dt, "render": render}


if __name__ == "__main__":
    main()
