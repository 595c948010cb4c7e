#!/usr/bin/env python3
"""The reference's FULL mapping run (BASELINE configs[1]: "Replica office0 full mapping loop") measured end to end.

What runs is the reference's own loop, through the mirrors of its callers:

    MappingModule.step(item | None)          /root/reference/nerf_vo/mapping/mapping_module.py:35-55
      -> Nerfstudio.__call__ -> update + train   /root/reference/nerf_vo/mapping/nerfstudio.py:111-173
         -> trainer.train_iteration(step)         (one replayed hipGraph per iteration)

with the reference's ingest CADENCE (`configs/nerf_vo_replica.yaml:14-24`: 192 keyframes, 8192 iterations): the queue
delivers one item per new keyframe, DPVO-shaped -- the new frame's colour plus the refreshed poses / depths of the
tracker's sliding window (`removal_window - 2` = 26 keyframes, /root/reference/nerf_vo/enhancement/
enhancement_module.py:32-37, ingest semantics /root/reference/nerf_vo/mapping/nerfstudio_utils.py:157-228) -- and
keyframe k arrives once k x `mapping_iterations / num_keyframes` (42.67) iterations are spent, i.e. the tracker sets the
pace and the mapper's idle-tick throttle (at most 42.67 training ticks between two items) is what fills the time between
arrivals; after the last frame it free-runs to 8192.  DEFAULT (non-deterministic) kernels, GradScaler regime, fixed exact poses.

Reported: wall seconds of the whole run (first item to iteration 8192, device-synchronised at both ends), ms per iteration
overall, the GPU time the ingests took (HIP events around `update`), timed windows of the loop AS IT RUNS (ingests
included) at iterations ~500 / 2000 / 5000 / 7900, and the per-kernel table of the step on the TRAINED field (eager
steps with a HIP-event pair per launch after the run; update and non-update steps in the late schedule's proportion).

    python tools/mapping_loop.py [--keyframes 192 --height 480 --width 640 --iterations 8192]
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

WINDOW_STARTS = (500, 2000, 5000, 7900)


def kernel_table(lib, engine, step_fn, steps: int, ms_per_step: float, device) -> list:
    """[(bench name, launches, total ms)] of ``steps`` EAGER steps with a HIP-event pair per launch (nvo_profile_*; the
    launchers' hooks only see eager launches).  Eager launches are host-bound, so a device-side spin of about one step
    at the head of every profiled step lets the host run ahead and the scopes see their kernels back to back."""
    spin = 0
    if hasattr(torch.cuda, "_sleep"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1_000_000)
        torch.cuda.synchronize(device)
        e0.record()
        torch.cuda._sleep(1_000_000)
        e1.record()
        torch.cuda.synchronize(device)
        spin = int(1_000_000 * min(2.0, max(0.3, 1.5 * ms_per_step)) / max(e0.elapsed_time(e1), 1e-3))
    cfg = engine.cfg
    saved = (cfg.overlap_proposal_backward, cfg.overlap_pose_backward)
    cfg.overlap_proposal_backward = cfg.overlap_pose_backward = False  # every kernel alone on the GPU
    lib.nvo_profile_enable(1)
    try:
        for _ in range(steps):
            if spin:
                torch.cuda._sleep(spin)
            step_fn()
        torch.cuda.synchronize(device)
        need = lib.nvo_profile_summary(None, 0)
        buf = C.create_string_buffer(int(need) + 16)
        lib.nvo_profile_summary(buf, len(buf))
    finally:
        lib.nvo_profile_enable(0)
        cfg.overlap_proposal_backward, cfg.overlap_pose_backward = saved
    rows = []
    for line in buf.value.decode().strip().splitlines():
        name, cnt, total = line.rsplit(",", 2)
        rows.append((name, int(cnt), float(total)))
    rows.sort(key=lambda r: -r[2])
    return rows


def run(keyframes=192, height=480, width=640, iterations=8192, window_steps=200, window_starts=WINDOW_STARTS,
        tracker_window=26, profile_steps=60, camera_optimizer_mode="off", seed=42, device="cuda:0", quiet=True,
        out_dir=None, render_frames=3):
    entry.build()
    from nerf_vo_amd import _lib
    from nerf_vo_amd.mapping.dataset import opencv_to_opengl
    from nerf_vo_amd.mapping.mapping_module import MappingModule
    from nerf_vo_amd.mapping.nerfstudio_mapper import Nerfstudio
    from nerf_vo_amd.synthetic import make_sequence

    torch.manual_seed(int(seed))
    dev = torch.device(device)
    out_dir = out_dir or tempfile.mkdtemp(prefix="nvo_loop_")
    args = argparse.Namespace(experiment="mapping_loop", dir_prediction=out_dir + "/pred", mapping_snapshot_iterations=iterations,
                              mapping_iterations=iterations, num_keyframes=keyframes, frame_height=height, frame_width=width,
                              enhancement_module="depth", deterministic=False, dynamic_loss_scale=None,
                              camera_optimizer_mode=camera_optimizer_mode)
    mapper = Nerfstudio(args, device=dev)
    module = MappingModule(mapper, mapping_iterations=iterations, num_keyframes=keyframes)
    eng = mapper.trainer.pipeline.model.engine
    seq = make_sequence(keyframes, height, width, device=dev)
    poses_gl = opencv_to_opengl(seq["camera_extrinsics"])

    def item(k: int) -> dict:
        """What the enhancement stage pushes when keyframe k arrives from a sparse (DPVO) tracker."""
        lo = max(0, k + 1 - tracker_window)
        return {"keyframe_indices": torch.arange(lo, k + 1), "camera_intrinsics": seq["camera_intrinsics"][k:k + 1],
                "camera_extrinsics": poses_gl[lo:k + 1], "frames_color": seq["frames_color"][k:k + 1],
                "frames_depth": seq["frames_depth"][lo:k + 1], "last_frame": k == keyframes - 1}

    # GPU time of the ingests: an event pair around every Nerfstudio.update
    ingest_events = []
    plain_update = mapper.update

    def timed_update(input):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plain_update(input=input)
        e1.record()
        ingest_events.append((e0, e1))

    mapper.update = timed_update

    starts = sorted(s for s in window_starts if s + window_steps <= iterations)
    windows, open_w = [], None
    updates_in_window = 0
    ticks = skipped = 0
    next_kf = 0

    def window_hooks():
        """Called before every tick: opens / closes the timed windows at iteration boundaries (device-synchronised)."""
        nonlocal open_w, updates_in_window
        if open_w is not None and mapper.step >= open_w["first_iteration"] + window_steps:
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - open_w.pop("t0")
            n = mapper.step - open_w["first_iteration"]
            open_w.update({"iterations": n, "ms_per_iteration": dt / n * 1e3, "ray_samples_per_sec": n * eng.cfg.num_rays * eng.cfg.num_nerf_samples / dt,
                           "ingests_inside": open_w["ingests_inside"], "proposal_updates": updates_in_window,
                           "keyframes_active_at_end": int(mapper.trainer.pipeline.datamanager.train_dataset.num_active_frames),
                           "loss_scale": eng.current_loss_scale(),
                           "sparse_steps": bool(getattr(eng, "_sparse_mode", False))})
            windows.append(open_w)
            open_w = None
        if open_w is None and starts and mapper.step >= starts[0]:
            starts.pop(0)
            torch.cuda.synchronize(dev)
            open_w = {"first_iteration": mapper.step, "ingests_inside": 0, "t0": time.perf_counter()}
            updates_in_window = 0

    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    while not module.shutdown and mapper.step < iterations:
        window_hooks()
        # keyframes arrive evenly over the run (the tracker sets the pace: keyframe k is in the queue once the mapper has
        # spent k x mapping_iterations / num_keyframes iterations -- the budget MappingModule.step's throttle allows per item)
        deliver = next_kf < keyframes and mapper.step >= next_kf * iterations / keyframes
        before = mapper.step
        if deliver:
            module.step(item(next_kf))
            next_kf += 1
            if open_w is not None:
                open_w["ingests_inside"] += 1
        else:
            _, skip = module.step(None)
            skipped += int(skip)
        ticks += 1
        if mapper.step > before and eng.steps_since_proposal_update == 1:
            updates_in_window += 1
    window_hooks()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    done = mapper.step
    ingest_ms = sum(e0.elapsed_time(e1) for e0, e1 in ingest_events)
    losses = eng.loss_dict()

    # ---- the step on the TRAINED field, kernel by kernel (eager, outside the timed run; the parameters keep training)
    ds = mapper.trainer.pipeline.datamanager.train_dataset
    dm = mapper.trainer.pipeline.datamanager
    c2w = ds.camera_extrinsics[:, :3, :4].contiguous()

    def eager_step():
        ray_indices, _ = dm.next_train(eng.step)
        eng.train_step(ray_indices, ds.camera_intrinsics, c2w, ds.frames_color, ds.frames_depth)

    table = []
    if profile_steps > 0:
        late_ms = windows[-1]["ms_per_iteration"] if windows else 0.5
        rows = kernel_table(_lib.lib(), eng, eager_step, profile_steps, late_ms, dev)
        tot = sum(r[2] for r in rows)
        for name, cnt, total in rows:
            table.append({"kernel": name, "launches_per_step": round(cnt / profile_steps, 3), "avg_launch_us": round(total / cnt * 1e3, 2),
                          "us_per_step": round(total / profile_steps * 1e3, 2), "share": round(total / tot, 4)})
    # (no shut_down tick: the snapshot -- a 0.9 GB dataset.pt at this size -- is outside the measured path)
    # ---- inference on the field this run trained (what evaluation/nerf_renderer.py does after mapping): full frames at the
    # dataset's native 1200x680 and at the training resolution, one hipGraph per image shape, HIP-event timed
    rendered = []
    if render_frames > 0:
        from nerf_vo_amd.mapping.cameras import Cameras
        from nerf_vo_amd.synthetic import replica_intrinsics

        pose = ds.camera_extrinsics[keyframes // 2:keyframes // 2 + 1, :3, :4].clone()
        for w_, h_ in ((1200, 680), (width, height)):
            fx, fy, cx, cy = replica_intrinsics(h_, w_)
            cams = Cameras(camera_to_worlds=pose, fx=fx, fy=fy, cx=cx, cy=cy, width=w_, height=h_).to(dev)

            def frame():
                b = cams.generate_rays(camera_indices=0, keep_shape=True)
                return eng.render_image(b.origins.reshape(-1, 3), b.directions.reshape(-1, 3),
                                        b.metadata["directions_norm"].reshape(-1), chunk=1 << 15)

            frame()  # (captures the graph of this image shape)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(render_frames):
                img = frame()
            e1.record()
            torch.cuda.synchronize(dev)
            rendered.append({"resolution": [w_, h_], "ms_per_frame": round(e0.elapsed_time(e1) / render_frames, 3),
                             "finite": bool(torch.isfinite(img["rgb"]).all()), "mean_accumulation": round(float(img["accumulation"].mean()), 4)})
    res = {
        "what": "the reference's mapping run end to end: MappingModule.step -> Nerfstudio(update | train) with its ingest cadence "
                "(one DPVO-shaped item per keyframe: new colour frame + refreshed poses / depths of the tracker window, then "
                "<= mapping_iterations / num_keyframes idle training ticks), default non-deterministic kernels, GradScaler regime",
        "keyframes": keyframes, "resolution": [width, height], "iterations": done, "tracker_window": tracker_window,
        "camera_optimizer_mode": camera_optimizer_mode, "wall_seconds": wall, "ms_per_iteration": wall / max(done, 1) * 1e3,
        "ray_samples_per_sec": done * eng.cfg.num_rays * eng.cfg.num_nerf_samples / wall,
        "ticks": ticks, "skipped_ticks": skipped, "ingests": len(ingest_events), "ingest_gpu_ms_total": ingest_ms,
        "ingest_gpu_ms_each": ingest_ms / max(len(ingest_events), 1),
        "windows": windows, "final_losses": losses, "loss_scale_end": eng.current_loss_scale(),
        # EngineConfig.sparse_backward: which kind of step the run ended on, and the live-tile fraction its
        # last probe read (one of a ray's three tiles on a trained field; 1.0 while GradScaler's scale keeps everything live)
        "sparse_backward": {"config": eng.cfg.sparse_backward, "sparse_steps_at_the_end": bool(getattr(eng, "_sparse_mode", False)),
                            "live_tile_fraction_last_probe": getattr(eng, "_sparse_live_frac", None)},
        "render_of_the_trained_field": rendered,
        "trained_field_kernel_table": table,
        "trained_field_kernel_us_per_step": round(sum(r["us_per_step"] for r in table), 2),
        "trained_field_note": f"{profile_steps} eager steps after iteration {done}, HIP-event pair per launch, every kernel alone on "
                              "the GPU; proposal networks refreshed every 6th step as the schedule has it there",
    }
    if not quiet:
        print(json.dumps(res))
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--keyframes", type=int, default=192)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--iterations", type=int, default=8192)
    ap.add_argument("--window-steps", type=int, default=200)
    ap.add_argument("--profile-steps", type=int, default=60)
    ap.add_argument("--camera-optimizer-mode", default="off", help="off (BASELINE configs[1], fixed poses) | SE3 | SO3xR3")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--render-frames", type=int, default=3, help="timed full-frame renders of the trained field per resolution (0 = skip)")
    a = ap.parse_args()
    run(a.keyframes, a.height, a.width, a.iterations, window_steps=a.window_steps, profile_steps=a.profile_steps,
        camera_optimizer_mode=a.camera_optimizer_mode, seed=a.seed, quiet=False, render_frames=a.render_frames)
