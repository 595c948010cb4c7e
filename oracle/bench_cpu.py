"""cpu_baseline leg of bench.py (TEST INFRASTRUCTURE): times the torch-CPU oracle of the SAME training
step (forward + losses + autograd backward + Adam on every parameter) on the GPU box's host cores.

The reference's own CPU path (nerfstudio ``implementation="torch"``) cannot be imported here
(SURVEY.md section 8c), so this is kind "port".  Two forms:
  * BASELINE.json configs[0] as written (bench.py default): ONE 640x480 keyframe, 4096 rays drawn from it, fp32,
    the MEDIAN of 20 timed steps after 3 warm-ups (SURVEY.md section 8d's protocol; about a minute).  Thread count: torch-CPU gets SLOWER past a few dozen threads on these
    gather/scatter-heavy ops (256 threads on the GPU box's host measured 40x slower than 16), so the count is
    calibrated on a 128-ray forward+backward (8, 16, 32, ... up to every host core, stopping at the first count that
    is slower) and the best one is used; both the threads used and the host's core count are reported.
  * an explicitly labelled 256-ray sample of the same step (8 timed steps, about 12 s).
Throughput is reported in the metric's unit (main-field ray-samples/s).
"""
from __future__ import annotations

import os
import time

import torch

from .nerfacto import NerfactoOracle, OracleConfig, adam_reference


def _calibrate_threads(orc, host_cores: int) -> tuple[int, list]:
    """Fastest torch thread count for this step on this host (see the module docstring)."""
    g = torch.Generator().manual_seed(1)
    R = 128
    origins = (torch.rand(R, 3, generator=g) - 0.5) * 0.8
    directions = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    cam = torch.zeros(R, dtype=torch.long)
    jit = tuple(torch.rand(R, generator=g) for _ in range(3))
    gt_rgb, gt_depth = torch.rand(R, 3, generator=g), torch.rand(R, generator=g) * 2
    cands = sorted({c for c in (8, 16, 32, 64, 128, 256, host_cores) if c <= host_cores} | {min(host_cores, 8)})
    best, best_t, log = cands[0], float("inf"), []
    for c in cands:
        torch.set_num_threads(c)
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            orc.zero_grad()
            out = orc.forward(origins, directions, torch.ones(R), cam, jit, anneal=1.0, training=True)
            sum(orc.loss_dict(out, gt_rgb, gt_depth).values()).backward()
            ts.append(time.perf_counter() - t0)
        log.append((c, round(min(ts), 3)))
        if min(ts) < best_t:
            best, best_t = c, min(ts)
        elif min(ts) > 1.1 * best_t:
            break
    return best, log


def time_cpu_step(num_rays: int = 256, num_images: int = 8, steps: int = 8, warmup: int = 1,
                  max_threads: int | None = 16, keyframe: tuple | None = None) -> dict:
    """keyframe=(H, W): BASELINE configs[0] -- the rays are pixels of ONE synthetic keyframe of that size (its colours
    and depths are the targets); max_threads=None: thread count calibrated on this host."""
    host_cores = os.cpu_count() or 1
    cfg = OracleConfig(num_images=num_images, emulate_fp16=False, dtype=torch.float32)
    orc = NerfactoOracle(cfg)
    orc.init_random(0)
    calib = None
    if max_threads is None:
        cores, calib = _calibrate_threads(orc, host_cores)
    else:
        # torch-CPU scales poorly past a few dozen threads on these small gather/scatter ops (256 threads
        # on the GPU box's host ran 40x SLOWER than 16); "cores" reports the threads actually used.
        cores = min(host_cores, max_threads)
    torch.set_num_threads(cores)
    state = {k: (torch.zeros_like(p), torch.zeros_like(p)) for k, p in orc.params.items()}
    g = torch.Generator().manual_seed(0)
    frame = None
    if keyframe is not None:
        from . import rays as Rr
        from nerf_vo_amd.synthetic import make_sequence  # data generator only (torch-CPU); no kernel code

        H, W = keyframe
        seq = make_sequence(1, H, W, device="cpu")
        c2w = seq["camera_extrinsics"][:, :3, :4].clone()
        c2w[:, :, 1:3] *= -1  # OpenCV -> OpenGL axes
        frame = (seq["camera_intrinsics"], c2w, seq["frames_color"][0].permute(1, 2, 0).contiguous(),
                 seq["frames_depth"][0, 0].contiguous(), Rr, H, W)

    def one_step(step_idx: int):
        R = num_rays
        if frame is not None:
            intr, c2w, color, depth, Rr, H, W = frame
            idx = torch.stack([torch.zeros(R, dtype=torch.long), torch.randint(0, H, (R,), generator=g),
                               torch.randint(0, W, (R,), generator=g)], dim=1)
            origins, directions, dnorm, _ = Rr.generate_rays(idx, intr, c2w)
            origins, directions, dnorm = origins.float(), directions.float(), dnorm.reshape(-1).float()
            cam = idx[:, 0]
            gt_rgb, gt_depth = color[idx[:, 1], idx[:, 2]], depth[idx[:, 1], idx[:, 2]]
        else:
            origins = (torch.rand(R, 3, generator=g) - 0.5) * 0.8
            directions = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
            dnorm = torch.ones(R)
            cam = torch.randint(0, num_images, (R,), generator=g)
            gt_rgb = torch.rand(R, 3, generator=g)
            gt_depth = torch.rand(R, generator=g) * 2
        jit = tuple(torch.rand(R, generator=g) for _ in range(3))
        orc.zero_grad()
        out = orc.forward(origins, directions, dnorm, cam, jit, anneal=1.0, training=True)
        loss = sum(orc.loss_dict(out, gt_rgb, gt_depth).values())
        loss.backward()
        with torch.no_grad():
            for k, p in orc.params.items():
                if p.grad is None:
                    continue
                m, v = state[k]
                new_p, m, v = adam_reference(p, p.grad, m, v, 1e-2, step_idx + 1)
                p.copy_(new_p)
                state[k] = (m, v)

    for i in range(warmup):
        one_step(i)
    per_step = []
    for i in range(steps):
        t0 = time.perf_counter()
        one_step(warmup + i)
        per_step.append(time.perf_counter() - t0)
    per_step.sort()
    dt = per_step[len(per_step) // 2]  # MEDIAN step (median of >= 20 steps after 3 warm-ups)
    mean_dt = sum(per_step) / len(per_step)
    samples = num_rays * cfg.num_nerf_samples
    what = (f"BASELINE configs[0]: one {keyframe[1]}x{keyframe[0]} keyframe, " if keyframe is not None else
            "256-ray SAMPLE of the step (fallback form), ")
    return {"value": samples / dt, "unit": "ray-samples/s", "cores": cores, "kind": "port",
            "sample": what + f"median of {steps} timed steps after {warmup} warm-ups x {num_rays} rays (256/96/48 samples per ray, "
                      f"full {sum(p.numel() for p in orc.params.values()) / 1e6:.1f} M-parameter model, float32 torch-CPU "
                      f"oracle incl. autograd backward + Adam), {dt:.2f} s/step (mean {mean_dt:.2f}), {cores} threads of "
                      f"{host_cores} host cores (calibrated: torch-CPU gets slower past a few dozen threads on these ops)",
            "seconds_per_step": dt, "seconds_per_step_mean": mean_dt, "steps": steps, "warmup": warmup,
            "host_cores": host_cores, "thread_calibration": calib}
