#!/usr/bin/env python3
"""One rank of the 2-process data-parallel check (tests/test_distributed_gpu.py): both ranks share cuda:0 and
exchange gradients over gloo (RCCL needs one device per rank; the code path above the collective is the same).
Usage: dist_engine_worker.py <rank> <world> <port> <workdir> <compress: none|fp16>"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

entry.build()
from nerf_vo_amd.engine import EngineConfig, NerfactoEngine  # noqa: E402
from nerf_vo_amd.mapping.dataset import DynamicDataset, opencv_to_opengl  # noqa: E402
from nerf_vo_amd.parallel import GradientAllReduce  # noqa: E402
from nerf_vo_amd.synthetic import make_sequence  # noqa: E402


def main():
    rank, world, port, workdir, compress = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    plan = torch.load(os.path.join(workdir, "plan.pt"))
    n, H, W, R = plan["n"], plan["H"], plan["W"], plan["R"]
    normals = bool(plan.get("normals", False))  # BASELINE configs[4]: monosdf normal supervision (+ bf16 MLPs)
    ds = DynamicDataset(num_frames=n, frame_height=H, frame_width=W, device=dev, use_normals=normals)
    seq = make_sequence(n, H, W, device=dev)
    ds.update({"keyframe_indices": torch.arange(n), "camera_intrinsics": seq["camera_intrinsics"],
               "camera_extrinsics": opencv_to_opengl(seq["camera_extrinsics"]), "frames_color": seq["frames_color"],
               "frames_depth": seq["frames_depth"], **({"frames_normal": seq["frames_normal"]} if normals else {})})
    torch.manual_seed(100)  # SAME seed on every rank: the engine itself must give each rank its own sampler stream
    eng = NerfactoEngine(EngineConfig(num_images=n, num_rays=R, optimize_poses=plan["poses"],
                                      pipeline_sampling_prefix=plan.get("pipeline", True),
                                      mlp_dtype=plan.get("mlp_dtype", "f16"), expect_normals=normals,
                                      **plan.get("engine_overrides", {})), dev,
                         world_size=world, rank=rank)
    eng.set_params(plan["params"].to(dev))  # identical initial parameters on every rank
    reducer = GradientAllReduce(dist, compress=None if compress == "none" else compress,
                                shard_optimizer=bool(plan.get("shard_optimizer", False)))
    c2w = ds.camera_extrinsics[:, :3, :4].contiguous()
    for k in range(plan["eager_steps"]):
        idx = plan["rays"][k][rank].to(dev)
        jit = tuple(j.to(dev) for j in plan["jitters"][k][rank])
        eng.train_step(idx, ds.camera_intrinsics, c2w, ds.frames_color, ds.frames_depth, jitters=jit, all_reduce=reducer,
                       normals=ds.world_normals01() if normals else None)
    torch.cuda.synchronize()
    after_eager = eng.params.detach().cpu().clone()
    prefix_mismatch, prefix_checked = [], 0
    for _ in range(plan["graph_steps"]):
        eng.train_step_graphed(ds, all_reduce=reducer)
        if plan.get("verify_prefix") and eng._pending_head is not None:
            # The sampling prefix of the NEXT step has been launched ahead of this step's fields reduction + Adam.  It
            # is deterministic in its inputs (no float atomics before the losses), so replaying it once more now --
            # after everything of this step has landed -- must reproduce every buffer it writes BIT FOR BIT; a prefix
            # that raced with the exchange or read a parameter the late optimiser graph still had to write would not.
            torch.cuda.synchronize()
            ws = eng._workspace(R, True)
            km = len(eng.prop_nets)
            names = ["origins", "directions", "gt_rgb", "gt_depth", "cam_idx", "sh"] + [
                f"{b}{k}" for k in range(km + 1) for b in ("x", "sbins", "tbins")] + [
                f"{b}{k}" for k in range(km) for b in ("out", "weights")]
            before = {k: ws[k].clone() for k in names}
            next(iter(eng._graphs.values()))["head"].replay()  # (the sampling prefix is the same in every step variant)
            torch.cuda.synchronize()
            prefix_checked += 1
            prefix_mismatch += [k for k in names if not torch.equal(before[k].view(torch.uint8), ws[k].view(torch.uint8))]
    torch.cuda.synchronize()
    extra = {}
    if plan.get("checkpoint_render"):
        # NO manual sync_sharded_state() before either: the render reads the all-gathered working copy, state_dict gathers
        from nerf_vo_amd.mapping.model import ExtendedNerfactoModel

        g = torch.Generator().manual_seed(5)
        o = ((torch.rand(256, 3, generator=g) - 0.5) * 0.6).to(dev)
        d = torch.nn.functional.normalize(torch.randn(256, 3, generator=g), dim=-1).to(dev)
        extra["render_before_sync"] = eng.render_rays(o, d, torch.ones(256, device=dev))["rgb"].cpu().clone()
        model = object.__new__(ExtendedNerfactoModel)
        model.engine = eng
        extra["checkpoint"] = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in model.state_dict(all_reduce=reducer).items()}
        torch.cuda.synchronize()
    sharded_state = None
    if plan.get("shard_optimizer"):
        # sharded optimiser: the fp32 master / moments of the fields group are current on their owner only, the
        # 16-bit working copy everywhere -- record both views, then gather the fp32 state
        sharded_state = {"params_half": eng.params_half.cpu().clone(), "own_master": eng.params.cpu().clone()}
        eng.sync_sharded_state(reducer)
        torch.cuda.synchronize()
        sharded_state["captured_collectives"] = bool(next(iter(eng._graphs.values())).get("captured_collectives")) if eng._graphs else False
    ab = None
    if plan.get("ab_pipeline"):
        # A/B from ONE bit-identical state (the graphs of both step kinds exist by now): the same few steps with the
        # sampling prefix launched ahead, and twice in program order (the second gives the run-to-run noise of the
        # step's float atomics).  Starting from identical parameters / moments / counters keeps chaotic amplification
        # out of the comparison, which 12-step trajectories from separate processes could not.
        snap = ([t.clone() for t in (eng.params, eng.params_half, eng.exp_avg, eng.exp_avg_sq)], dict(eng.opt_steps),
                eng.step, eng.steps_since_proposal_update)

        def run(pipeline: bool):
            for dst, src in zip((eng.params, eng.params_half, eng.exp_avg, eng.exp_avg_sq), snap[0]):
                dst.copy_(src)
            eng.opt_steps = snap[1]
            eng.step, eng.steps_since_proposal_update, eng._pending_head = snap[2], snap[3], None
            eng.cfg.pipeline_sampling_prefix = pipeline
            for _ in range(plan["ab_pipeline"]):
                eng.train_step_graphed(ds, all_reduce=reducer)
            torch.cuda.synchronize()
            return (eng.params.detach() - snap[0][0]).double().cpu()

        ab = {"pipe": run(True), "serial": run(False), "serial2": run(False)}
    drawn = next(iter(eng._graphs.values()))["buffers"][1].cpu()  # pixel indices of the last graph-replayed step
    torch.save({"after_eager": after_eager, "after_graph": eng.params.detach().cpu(), "losses": eng.loss_dict(),
                "skip": eng.skip_flag.cpu(), "ray_indices": drawn, "prefix_checked": prefix_checked,
                "prefix_mismatch": sorted(set(prefix_mismatch)), "ab": ab, "sharded_state": sharded_state,
                "opt_steps": eng.opt_steps, "exp_avg": eng.exp_avg.cpu(), **extra}, os.path.join(workdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
