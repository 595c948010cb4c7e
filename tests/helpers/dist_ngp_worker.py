#!/usr/bin/env python3
"""One rank of the 2-process check of the occupancy-grid ("instant-ngp") back-end (tests/test_distributed_gpu.py):
both ranks share cuda:0 and exchange over gloo.  Every rank evaluates the density network at its OWN jittered point per
grid cell; the fresh estimates are MAX-reduced (parallel.GradientAllReduce.reduce_max) before the EMA, so the density
grid -- hence the Morton bitfield every rank marches through -- and, with the summed gradients, the parameters must
stay BIT-identical on all ranks.
Usage: dist_ngp_worker.py <rank> <world> <port> <workdir> <reduce: 1|0>"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

entry.build()
from nerf_vo_amd.mapping.dataset import opencv_to_opengl  # noqa: E402
from nerf_vo_amd.ngp_engine import NgpConfig, NgpEngine  # noqa: E402
from nerf_vo_amd.parallel import GradientAllReduce  # noqa: E402
from nerf_vo_amd.synthetic import make_sequence  # noqa: E402


def main():
    rank, world, port, workdir, use_reduce = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5] == "1"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    plan = torch.load(os.path.join(workdir, "plan.pt"))
    n, H, W, R, steps = plan["n"], plan["H"], plan["W"], plan["R"], plan["steps"]
    seq = make_sequence(n, H, W, device=dev, scene_scale=0.2)
    c2w = opencv_to_opengl(seq["camera_extrinsics"])
    c2w[:, :3, 3] += 0.5  # centre the room on the unit cube of cascade 0
    c2w = c2w[:, :3, :4].contiguous()
    images = seq["frames_color"].permute(0, 2, 3, 1).contiguous()
    depths = seq["frames_depth"].permute(0, 2, 3, 1).contiguous()
    eng = NgpEngine(NgpConfig(num_images=n, num_rays=R, capacity=1 << 16, adaptive_rays=False,
                              density_update_every=plan["update_every"]), dev, world_size=world)
    eng.set_params(plan["params"].to(dev))  # identical initial parameters on every rank
    reducer = GradientAllReduce(dist) if use_reduce else None
    torch.manual_seed(1000 + rank)  # every rank: its own cell jitters, rays and ray jitters
    scale = torch.tensor([n, H, W], device=dev)
    grids, bits = [], []
    for it in range(steps):
        idx = torch.floor(torch.rand(R, 3, device=dev) * scale).long()
        eng.train_step(idx, seq["camera_intrinsics"], c2w, images, depths, all_reduce=reducer)
        if it % plan["update_every"] == 0:
            grids.append(eng.density_grid.cpu().clone())
            bits.append(eng.bitfield.cpu().clone())
    torch.cuda.synchronize()
    torch.save({"grids": grids, "bits": bits, "params": eng.params.cpu(), "pose": eng.pose_adjustment.cpu(),
                "losses": eng.loss_dict(), "skip": eng.skip_flag.cpu(), "rays": idx.cpu()},
               os.path.join(workdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
