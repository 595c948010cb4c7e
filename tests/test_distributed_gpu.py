"""Data-parallel mapping step with TWO processes: rays sharded across ranks, one gradient
exchange per iteration, replicated optimiser.  Both ranks share the single GPU of the test box and exchange over gloo
(RCCL wants one device per rank); everything above the collective -- per-rank sampling, loss normalisation by the
global ray count, fp16-compressed flat gradient, split graph replay, per-group skip flags -- is the production path.

Checked: (1) parameters stay bit-identical across the ranks through eager and graph-replayed steps; (2) after the
eager steps they equal a single-process run fed the concatenated ray batch, up to summation order (uncompressed
exchange: 5e-3 relative L1 on the parameter update; bf16-compressed: 2e-2)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(workdir, world, compress):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "dist_engine_worker.py"), str(r),
                               str(world), str(port), str(workdir), compress], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-2000:] for o in outs)
    return [torch.load(workdir / f"rank{r}.pt") for r in range(world)]


@pytest.mark.parametrize("poses", [False, True], ids=["fixed-poses", "pose-optimisation"])
def test_pipelined_sampling_prefix_keeps_the_trajectory(device, tmp_path, poses):
    """The multi-GPU step launches the NEXT iteration's sampling prefix (rays -> proposal sampling) while the fields
    gradient of the current one is still in the collective, and reduces / steps the small groups (proposal networks,
    camera poses) first.  The reordering must not change what is computed.  The step is not bitwise reproducible run
    to run (float atomics in the MLP weight-gradient flush and in multi-chunk grid slices), so the yardstick is the
    run-to-run noise itself: 12 graph-replayed steps (update and non-update iterations, graph switches) in program
    order TWICE give the noise floor; the pipelined run must sit within 3x of it (a prefix that read a stale parameter
    would draw different samples -- orders of magnitude above the floor)."""
    from nerf_vo_amd.engine import EngineConfig, NerfactoEngine

    n, H, W, R, world = 6, 60, 80, 512, 2
    ref = NerfactoEngine(EngineConfig(num_images=n, num_rays=R, optimize_poses=poses), device)
    params0 = ref.params.detach().cpu().clone()
    del ref
    res = {}
    for tag, pipeline in (("pipe", True), ("serial", False), ("serial2", False)):
        wd = tmp_path / tag
        wd.mkdir()
        torch.save({"n": n, "H": H, "W": W, "R": R, "params": params0, "rays": [], "jitters": [], "poses": poses,
                    "eager_steps": 0, "graph_steps": 12, "pipeline": pipeline}, wd / "plan.pt")
        res[tag] = _run_ranks(wd, world, "bf16")
    for tag, r in res.items():
        assert torch.equal(r[0]["after_graph"], r[1]["after_graph"]), f"ranks diverged ({tag})"
        assert int(r[0]["skip"].sum()) == 0 and not torch.equal(r[0]["after_graph"], params0)
    upd = {tag: (r[0]["after_graph"] - params0).double() for tag, r in res.items()}
    scale = float(upd["serial"].abs().sum())
    noise = float((upd["serial2"] - upd["serial"]).abs().sum()) / scale
    diff = float((upd["pipe"] - upd["serial"]).abs().sum()) / scale
    print(f"relative L1 of the 12-step update: run-to-run {noise:.3e}, pipelined vs program order {diff:.3e}")
    assert diff <= 3.0 * noise + 1e-7, f"pipelined prefix changed the trajectory: {diff:.3e} vs noise floor {noise:.3e}"
    for k, v in res["serial"][0]["losses"].items():
        assert abs(res["pipe"][0]["losses"][k] - v) <= 0.05 * abs(v) + 1e-9, (k, res["pipe"][0]["losses"][k], v)


@pytest.mark.parametrize("compress,poses", [("none", False), ("bf16", False), ("bf16", True), ("fp16", False)],
                         ids=["none", "bf16", "bf16-pose-optimisation", "fp16"])
def test_two_rank_step_matches_concatenated_batch(device, tmp_path, compress, poses):
    from nerf_vo_amd.engine import EngineConfig, NerfactoEngine
    from nerf_vo_amd.mapping.dataset import DynamicDataset, opencv_to_opengl
    from nerf_vo_amd.synthetic import make_sequence

    n, H, W, R, world, eager_steps, graph_steps = 6, 60, 80, 512, 2, 3, 4
    g = torch.Generator().manual_seed(77)
    # the concatenated batch, one process
    ref = NerfactoEngine(EngineConfig(num_images=n, num_rays=world * R, optimize_poses=poses), device)
    params0 = ref.params.detach().cpu().clone()
    scale = torch.tensor([n, H, W])
    rays = [[torch.floor(torch.rand(R, 3, generator=g) * scale).long() for _ in range(world)] for _ in range(eager_steps)]
    jitters = [[tuple(torch.rand(R, generator=g) for _ in range(3)) for _ in range(world)] for _ in range(eager_steps)]
    torch.save({"n": n, "H": H, "W": W, "R": R, "params": params0, "rays": rays, "jitters": jitters, "poses": poses,
                "eager_steps": eager_steps, "graph_steps": graph_steps}, tmp_path / "plan.pt")
    r0, r1 = _run_ranks(tmp_path, world, compress)
    # (1) replicated state never diverges
    assert torch.equal(r0["after_eager"], r1["after_eager"]) and torch.equal(r0["after_graph"], r1["after_graph"])
    assert not torch.equal(r0["after_eager"], params0) and not torch.equal(r0["after_graph"], r0["after_eager"])
    assert int(r0["skip"].sum()) == 0 and np.isfinite(list(r0["losses"].values())).all()
    # every process called torch.manual_seed with the SAME value: the ranks must still draw different rays (otherwise
    # the summed gradient is one rank's gradient and data parallelism silently adds nothing)
    same = float((r0["ray_indices"] == r1["ray_indices"]).all(dim=1).float().mean())
    assert same < 0.01, f"{same:.1%} of the two ranks' pixel samples coincide"
    # (2) same trajectory as ONE process training on the concatenated batch
    ds = DynamicDataset(num_frames=n, frame_height=H, frame_width=W, device=device, use_normals=False)
    seq = make_sequence(n, H, W, device=device)
    ds.update({"keyframe_indices": torch.arange(n), "camera_intrinsics": seq["camera_intrinsics"],
               "camera_extrinsics": opencv_to_opengl(seq["camera_extrinsics"]), "frames_color": seq["frames_color"],
               "frames_depth": seq["frames_depth"]})
    c2w = ds.camera_extrinsics[:, :3, :4].contiguous()
    for k in range(eager_steps):
        idx = torch.cat(rays[k]).to(device)
        jit = tuple(torch.cat([jitters[k][r][j] for r in range(world)]).to(device) for j in range(3))
        ref.train_step(idx, ds.camera_intrinsics, c2w, ds.frames_color, ds.frames_depth, jitters=jit)
    torch.cuda.synchronize()
    upd_ref = (ref.params.detach().cpu() - params0).double()
    upd_two = (r0["after_eager"] - params0).double()
    moved = upd_ref.abs() > 1e-6
    assert float(moved.float().mean()) > 0.01
    err = (upd_two - upd_ref).abs()
    # Adam normalises the update: entries whose gradient is tiny flip with summation order, so compare in aggregate
    rel = float(err[moved].sum() / upd_ref[moved].abs().sum())
    print(f"compress={compress}: relative L1 difference of the parameter update {rel:.3e}")
    # uncompressed: summation order only.  bf16 keeps every gradient's sign and magnitude to 8 bits (what Adam's
    # normalised update needs).  fp16 flushes the tiny gradients of rarely hit grid entries to zero, which Adam would
    # have turned into full-size steps: kept as an option, measured here, NOT what bench.py uses.
    tol = {"none": 5e-3, "bf16": 2e-2, "fp16": 0.5}[compress]  # measured: 2.9e-3 / 6.2e-3 / 1.95e-1
    assert rel < tol, f"two-rank update differs from the concatenated-batch update by {rel:.3e} (relative L1)"
