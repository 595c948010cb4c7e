"""cpu_baseline leg of bench.py (TEST INFRASTRUCTURE): times the torch-CPU oracle of the SAME training
step (forward + losses + autograd backward + Adam on every parameter) on the GPU box's host cores.

The reference's own CPU path (nerfstudio ``implementation="torch"``) cannot be imported here
(SURVEY.md section 8c), so this is kind "port".  The sample is bounded: a reduced ray batch (256 rays) for
8 timed steps after one warm-up (about 12 s of CPU work at ~1.4 s/step); throughput is reported in the metric's unit
(main-field ray-samples/s).  float32, all host threads.
"""
from __future__ import annotations

import os
import time

import torch

from .nerfacto import NerfactoOracle, OracleConfig, adam_reference


def time_cpu_step(num_rays: int = 256, num_images: int = 8, steps: int = 8, warmup: int = 1,
                  max_threads: int = 16) -> dict:
    # torch-CPU scales poorly past a few dozen threads on these small gather/scatter ops (256 threads
    # on the GPU box's host ran 40x SLOWER than 16); "cores" reports the threads actually used.
    cores = min(os.cpu_count() or 1, max_threads)
    torch.set_num_threads(cores)
    cfg = OracleConfig(num_images=num_images, emulate_fp16=False, dtype=torch.float32)
    orc = NerfactoOracle(cfg)
    orc.init_random(0)
    state = {k: (torch.zeros_like(p), torch.zeros_like(p)) for k, p in orc.params.items()}
    g = torch.Generator().manual_seed(0)

    def one_step(step_idx: int):
        R = num_rays
        origins = (torch.rand(R, 3, generator=g) - 0.5) * 0.8
        directions = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
        dnorm = torch.ones(R)
        cam = torch.randint(0, num_images, (R,), generator=g)
        jit = tuple(torch.rand(R, generator=g) for _ in range(3))
        gt_rgb = torch.rand(R, 3, generator=g)
        gt_depth = torch.rand(R, generator=g) * 2
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
    t0 = time.perf_counter()
    for i in range(steps):
        one_step(warmup + i)
    dt = (time.perf_counter() - t0) / steps
    samples = num_rays * cfg.num_nerf_samples
    return {"value": samples / dt, "unit": "ray-samples/s", "cores": cores, "kind": "port",
            "sample": f"{steps} steps x {num_rays} rays (256/96/48 samples per ray, full 13.8 M-parameter model, "
                      f"float32 torch-CPU oracle incl. autograd backward + Adam), {dt:.2f} s/step",
            "seconds_per_step": dt}
