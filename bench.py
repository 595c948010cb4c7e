#!/usr/bin/env python3
"""Benchmark of the mapping hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: under torch.distributed.run, or plainly -- it then
                                                          starts its N ranks itself, launch_ranks())

A "step" is ONE full depth-nerfacto training iteration as NeRF-VO's mapping runs it
(/root/reference/nerf_vo/mapping/nerfstudio.py:151): pixel sampling -> rays -> proposal sampling
(256 -> 96 -> 48) -> hash grid + fused MLPs -> compositing + rgb/interlevel/distortion/depth losses
-> backward -> Adam, on 4096 rays per GPU drawn from a resident synthetic Replica-shaped buffer of
192 keyframes at 640x480 (BASELINE.json configs[1], fixed poses).  Metric: training ray-samples/s =
main-field samples whose forward+backward+optimiser update completed per second, whole job.
Rays shard across ranks (weak scaling); the exchange per step is the gradient of the flat parameter
buffer over RCCL: a reduce-scatter of the fields group (Adam on the rank's 1/W slice, all-gather of the
16-bit working copy) plus, on update steps, an all-reduce of the proposal / pose ranges
(nerf_vo_amd/parallel.py; NVO_SHARD_OPT=0: one all-reduce, replicated Adam).

Extra JSON objects: "roofline" (dominant kernel, live HIP-event timing) and "cpu_baseline" (the
torch-CPU oracle of the same step on a bounded sample, rank 0 / N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import shutil
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md, L2 (per XCD): 4 MiB per XCD, 32 MiB aggregate, ~34.5 TB/s

WORKLOADS = {
    "replica": dict(keyframes=192, height=480, width=640, mlp_dtype="f16", normals=False,
                    name="BASELINE configs[1]: Replica-shaped full mapping step, depth-nerfacto (proposal sampling "
                         "256/96/48), fixed poses"),
    "replica360": dict(keyframes=192, height=360, width=640, mlp_dtype="f16", normals=False,
                       name="BASELINE configs[1] at the reference's Replica training resolution 360x640 "
                            "(configs/nerf_vo_replica.yaml:16-17)"),
    "scannet": dict(keyframes=512, height=240, width=320, mlp_dtype="bf16", normals=True,
                    name="BASELINE configs[4] (one GPU of it): ScanNet-shaped mapping step, 512 keyframes 240x320 "
                         "(configs/nerf_vo_scannet.yaml:15-17), depth + monosdf normal supervision, bf16 MFMA MLPs + "
                         "fp32 hash accumulate"),
}


def algorithmic_bytes(name: str, cfg, R: int, fused_adam_params: int = 0) -> float | None:
    """ALGORITHMIC bytes one launch of the named kernel moves (DESIGN.md section 'Kernels'; per-unit
    figures follow SURVEY.md section 8d)."""
    S = {"L16": cfg.num_nerf_samples}
    n_main = R * cfg.num_nerf_samples
    n_p0 = R * cfg.num_proposal_samples[0]
    n_p1 = R * cfg.num_proposal_samples[1]
    # per sample: 8 corners x F=2 x fp16 per level gathered (+ x 12 B in, 4 B/level out)
    if name == "grid_fwd[L16]":
        return n_main * (16 * 8 * 4 + 12 + 16 * 4)
    if name == "grid_fwd[L5]":
        return (n_p0 + n_p1) / 2 * (5 * 8 * 4 + 12 + 5 * 4)
    bwd_kinds = ("grid_bwd_lds", "grid_bwd_atomic", "grid_bwd_binned", "grid_bwd_stream")
    if name in tuple(k + "[L16]" for k in bwd_kinds):
        # scatter RMW 2 x 512 B (16 levels x 8 corners x 4 B, read + write) + 12 B position
        # + 64 B d(encoded) = 1100 B / sample.  (The kernels accumulate fp32 pairs -- twice the RMW bytes of the
        # fp16 table 8d prices -- which the contract does not credit: see roofline.algorithmic_bytes_fp32_rmw.)
        b = n_main * (16 * 8 * 4 * 2 + 12 + 16 * 4)
        if name == "grid_bwd_stream[L16]" and fused_adam_params:
            # the pass also takes the Adam step of the hashed levels' entries (EngineConfig.fuse_grid_adam): SURVEY 8d
            # prices Adam at 28 B per parameter (the gradient it no longer reads is part of that figure; not deducted)
            b += 28 * fused_adam_params
        return b
    if name in tuple(k + "[L5]" for k in bwd_kinds):  # one launch per proposal net: average of the two
        return (n_p0 + n_p1) / 2 * (5 * 8 * 4 * 2 + 12 + 5 * 4)
    if name == "mlp_bwd[64-64x2-16]":
        # drgb, rgb, 2 hidden, base_out in; d_base_out out (fp16 rows)
        return n_main * 2 * (16 + 16 + 128 + 16 + 16)
    if name == "mlp_fwd[64-64x2-16]":
        return n_main * 2 * (16 + 128 + 16)
    if name == "mlp_bwd[32-64x1-16]":
        return n_main * 2 * (16 + 16 + 64 + 32 + 32)
    if name == "mlp_fwd[32-64x1-16]":
        return n_main * 2 * (32 + 64 + 16)
    if name == "adam":
        return None  # several launches of different sizes: priced from the parameter count below
    return None


def pmc_traffic(name: str, live_path: str | None = None):
    """HBM bytes per launch of the named kernel from the committed rocprofv3 PMC summary
    (profiles/*_pmc_fetch_write_per_kernel.json: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of
    this same command; (FETCH_SIZE corrected for gfx950's half-counted coalesced streaming reads + WRITE_SIZE) *
    1024 -- see the note in that file).  The counters cannot be collected inside a timed run (a --pmc pass
    serialises the kernels), so the figure comes from the latest committed pass of the same command and names its
    source file.  None when no PMC summary covers the kernel."""
    import glob

    committed = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_fetch_write_per_kernel.json")), reverse=True)
    for path in ([live_path] if live_path else []) + committed:
        try:
            data = json.load(open(path))
        except Exception:
            continue
        for k in data.get("kernels", []):
            if k.get("bench_name") == name:
                fetch = k.get("FETCH_SIZE_KB_corrected", k["FETCH_SIZE_KB_per_launch"])  # gfx950 streaming-read correction
                raw = int((k["FETCH_SIZE_KB_per_launch"] + k["WRITE_SIZE_KB_per_launch"]) * 1024)
                src = ("live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this bench process" if path == live_path
                       else os.path.basename(path))
                return int((fetch + k["WRITE_SIZE_KB_per_launch"]) * 1024), src, raw
    return None, None, None


def workload_argv(args) -> list:
    """The flags that DEFINE the measured configuration (not what is reported about it), as a child bench process must
    receive them to measure the same thing."""
    out = ["--workload", args.workload, "--keyframes", str(args.keyframes), "--height", str(args.height), "--width", str(args.width),
           "--mlp-dtype", args.mlp_dtype, "--rays", str(args.rays)]
    for flag in ("optimize_poses", "static_loss_scale", "no_fuse_grid_adam", "no_graph", "no_overlap", "no_pose_overlap",
                 "commit_in_graph", "commit_behind_replay", "pipeline_single_gpu"):
        if getattr(args, flag, False):
            out.append("--" + flag.replace("_", "-"))
    if args.grid_bwd_mode is not None:
        out += ["--grid-bwd-mode", *[str(m) for m in args.grid_bwd_mode]]
    return out


def gather_roofline(kernel_table: list, cfg, R: int, summary_path: str | None):
    """`roofline_gather`: the main grid's hash gather (north_star: "rocprof HBM GB/s on the hash gather") priced three
    ways per launch -- the ALGORITHMIC gather bytes (588 B per main-field sample), the HBM traffic the PMC
    counters saw (FETCH_SIZE + WRITE_SIZE of the same launch: small, the 24 MB table lives in the L2s), and the read
    REQUESTS the vector L1s sent to the L2 (TCP_TCC_READ_REQ; each asks for one cache line -- 64 B counted, a 128-byte
    line filled): the figure the kernel is really bound by.  None without a kernel table."""
    row = next((r for r in kernel_table if r[0] == "grid_fwd[L16]"), None)
    if row is None:
        return None
    avg_s = row[2] / row[1] * 1e-3
    alg = algorithmic_bytes("grid_fwd[L16]", cfg, R)
    out = {"kernel": "grid_fwd[L16]", "avg_launch_us": round(avg_s * 1e6, 2), "algorithmic_bytes_per_launch": int(alg),
           "algorithmic": {"achieved": round(alg / avg_s / 1e9, 1), "unit": "GB/s", "frac_of_hbm_peak": round(alg / avg_s / 1e9 / HBM_PEAK_GBS, 4),
                           "frac_of_l2_peak": round(alg / avg_s / 1e9 / L2_PEAK_GBS, 4)},
           "hbm": None, "l2_requests": None}
    paths = ([summary_path] if summary_path else []) + sorted(
        __import__("glob").glob(os.path.join(ROOT, "profiles", "*_pmc_fetch_write_per_kernel.json")), reverse=True)
    for path in paths:
        try:
            data = json.load(open(path))
        except Exception:
            continue
        k = next((k for k in data.get("kernels", []) if k.get("kernel") == "k_grid_fwd"), None)
        if k is None:
            continue
        src = "live passes of this bench process" if path == summary_path else os.path.basename(path)
        hbm = int((k["FETCH_SIZE_KB_per_launch"] + k["WRITE_SIZE_KB_per_launch"]) * 1024)
        out["hbm"] = {"traffic_bytes_per_launch": hbm, "achieved": round(hbm / avg_s / 1e9, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
                      "frac": round(hbm / avg_s / 1e9 / HBM_PEAK_GBS, 4), "source": src,
                      "note": "raw FETCH_SIZE + WRITE_SIZE (gathers: no streaming-read correction); the table is L2-resident, "
                              "most of this is the positions read and the encoded features written"}
        if "L2_READ_REQ_per_launch" in k:
            req = k["L2_READ_REQ_per_launch"]
            out["l2_requests"] = {"read_requests_per_launch": int(req), "requests_per_sample_level": round(req / (R * cfg.num_nerf_samples * 16), 2),
                                  "bytes_at_64": int(req * 64), "bytes_at_128": int(req * 128), "peak": L2_PEAK_GBS, "unit": "GB/s",
                                  "achieved_at_64": round(req * 64 / avg_s / 1e9, 1), "achieved_at_128": round(req * 128 / avg_s / 1e9, 1),
                                  "frac_at_64": round(req * 64 / avg_s / 1e9 / L2_PEAK_GBS, 4),
                                  "frac_at_128": round(req * 128 / avg_s / 1e9 / L2_PEAK_GBS, 4),
                                  "amplification_over_algorithmic": [round(req * 64 / alg, 2), round(req * 128 / alg, 2)], "source": src}
        break
    return out


def live_pmc_summary(config_argv: list, steps: int = 20, warmup: int = 5, timeout_s: int = 240):
    """HBM traffic counters of THIS build in THIS session: two child runs of the bench command under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, --kernel-trace only, as
    MI355X_MICROARCH.md's HBM section prescribes; the counters serialise the kernels, so they cannot ride in the timed
    run), summarised by tools/pmc_traffic.py.  Children, never an exec: this process has initialised the GPU.  Returns
    (path of the summary JSON, None) or (None, reason)."""
    import shutil
    import subprocess
    import tempfile

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    work = tempfile.mkdtemp(prefix="nvo_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        rev = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        rev = ""
    env["NVO_COMMIT"] = (rev or "unknown") + " (collected by the bench run that printed the line)"
    # the child measures the SAME configuration as this process (workload, dtype, rays, pose optimisation, loss scale,
    # optimiser form ...): only the reporting sections are switched off
    cmd_tail = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", str(warmup), *config_argv,
                "--psnr", "off", "--cpu-baseline", "off", "--late-steps", "0", "--no-kernel-table", "--render-frames", "0",
                "--ngp-steps", "0", "--mapping-loop", "off", "--pmc-traffic", "off"]
    import signal

    # (third pass: read requests of the vector L1s to the L2 -- what the hash gathers are served by, roofline_gather)
    for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write"), ("TCP_TCC_READ_REQ_sum", "l2req")):
        cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(work, sub), "--", *cmd_tail]
        try:
            # own process group: on a timeout the WHOLE group goes (rocprofv3 is a wrapper; killing it alone would leave
            # its python child on the GPU)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                                    start_new_session=True)
        except OSError as exc:
            return None, f"{counter} pass: {type(exc).__name__}"
        optional = sub == "l2req"  # (the HBM counters are the contract; the L2 request pass only feeds roofline_gather)
        try:
            _, err = proc.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)  # (the exact group this call started)
            except OSError:
                pass
            proc.communicate()
            if optional:
                shutil.rmtree(os.path.join(work, sub), ignore_errors=True)
                continue
            return None, f"{counter} pass: timed out after {timeout_s} s (process group killed)"
        if proc.returncode != 0:
            if optional:
                shutil.rmtree(os.path.join(work, sub), ignore_errors=True)
                continue
            return None, f"{counter} pass exited {proc.returncode}: {(err or '')[-300:]}"
    out = os.path.join(work, "pmc_fetch_write_per_kernel.json")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(work, "fetch"),
                          os.path.join(work, "write"), out, str(steps + warmup), os.path.join(work, "l2req")], env=env,
                         capture_output=True, text=True)
    if res.returncode != 0 or not os.path.exists(out):
        return None, f"pmc_traffic.py: {res.stderr[-300:]}"
    try:  # the summary names the configuration it was collected on
        data = json.load(open(out))
        data["bench_argv"] = config_argv
        json.dump(data, open(out, "w"), indent=1)
    except (OSError, ValueError):
        pass
    try:  # keep the summary next to the run's other outputs (copied to profiles/ by hand when it is to be judged)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        shutil.copy(out, os.path.join(ROOT, "gpurun_out", "live_pmc_fetch_write_per_kernel.json"))
    except OSError:
        pass
    return out, None


def launch_ranks(n: int, argv: list) -> int:
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as a child job
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`,
    one rank per GPU), relay rank 0's single JSON line to stdout, return the job's exit code.  Runs BEFORE anything
    initialises the GPU in this process (`import torch` and device_count() do not), and nothing is exec'd: the ranks
    are children.  NVO_BENCH_DRY=1 (tests): the ranks only rendezvous over gloo and print a stub line."""
    import socket
    import subprocess

    dry = os.environ.get("NVO_BENCH_DRY") == "1"
    if not dry:
        have = torch.cuda.device_count()  # (counts devices without creating a context on this image)
        if have < n:
            sys.stderr.write(f"bench.py: --gpus {n} asked for, this node exposes {have} GPU(s): nothing was run "
                             f"(error: not-enough-gpus)\n")
            return 3
    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: RCCL across processes needs it on this pool)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    sys.stderr.write(f"[bench] launching {n} ranks: {' '.join(cmd)}\n")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:  # rank 0 prints exactly one JSON line; anything else on the ranks' stdout goes to stderr
        text = out.strip()
        if text.startswith("{") and text.endswith("}") and line is None:
            try:
                json.loads(text)
                line = text
                continue
            except ValueError:
                pass
        sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited cleanly without a result line (error: no-result)\n")
        rc = 4
    return rc


def dry_run(args) -> int:
    """NVO_BENCH_DRY=1: the launch path without a GPU -- rendezvous over gloo, the max-over-ranks timing reduction,
    rank 0's single line (tests/test_parallel_cpu.py runs `python bench.py --gpus 2` this way)."""
    import torch.distributed as dist

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: WORLD_SIZE {world} != --gpus {args.gpus}\n")
        return 2
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "training ray-samples/sec", "dry_run": True, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "max_over_ranks": float(t.item())}), flush=True)
    dist.destroy_process_group()
    return 0


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=tuple(WORKLOADS), default="replica",
                    help="replica = BASELINE configs[1] (192 keyframes 640x480, fp16 MLPs); replica360 = the same at the "
                         "reference's own Replica training resolution 360x640 (configs/nerf_vo_replica.yaml:16-17); scannet = "
                         "BASELINE configs[4] (512 keyframes 240x320 per configs/nerf_vo_scannet.yaml:15-17, depth + normal "
                         "supervision, bf16 MFMA MLPs + fp32 hash accumulate)")
    ap.add_argument("--keyframes", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--mlp-dtype", choices=("f16", "bf16"), default=None)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--cpu-baseline", choices=("configs0", "sample", "off"), default="configs0",
                    help="configs0 = BASELINE configs[0] as written: one 640x480 keyframe, 4096 rays, fp32 torch-CPU, calibrated "
                         "thread count, median of 20 timed steps after 3 warm-ups (about 1 min of CPU); sample = 256-ray sample (about 12 s)")
    ap.add_argument("--cpu-baseline-rays", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--psnr", choices=("off", "small", "replica"), default="replica",
                    help="second half of the BASELINE metric (render PSNR): an end-to-end run of the mapper mirror on "
                         "the synthetic room, held-out views rendered by the native renderer.  small = 48 keyframes "
                         "320x240, 1500 iterations (~5 s); replica = 192 keyframes 640x480, 8192 iterations (~10 s on MI355X)")
    ap.add_argument("--grid-bwd-mode", type=int, nargs="+", default=None,
                    help="hash-grid backward kernel: one value for all networks or three (main, proposal 0, "
                         "proposal 1); 3 = streamed binned, 1 = LDS slice owner, 2 = binned, 0 = global atomics; "
                         "default = EngineConfig's")
    ap.add_argument("--optimize-poses", action="store_true",
                    help="BASELINE configs[2]: SE3 pose-gradient backprop enabled (default: configs[1], fixed poses)")
    ap.add_argument("--late-steps", type=int, default=100,
                    help="extra timed steps late in the proposal-update schedule (reported as late_schedule; 0 = skip)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-overlap", action="store_true",
                    help="proposal backward on the main stream instead of a side stream (profiling: every kernel alone)")
    ap.add_argument("--no-kernel-table", action="store_true",
                    help="skip the eager per-kernel HIP-event pass (rocprofv3 runs: only graph-replayed steps in the trace)")
    ap.add_argument("--no-pose-overlap", action="store_true",
                    help="pose optimisation: the main grid's parameter scatter behind the pose chain instead of beside it (A/B)")
    ap.add_argument("--commit-in-graph", action="store_true",
                    help="the optimiser's commit as the graph's last node, scalars written eagerly BEFORE each replay (A/B)")
    ap.add_argument("--no-fuse-grid-adam", action="store_true",
                    help="Adam of the main grid's hashed levels in the optimiser launch instead of inside the grid backward (A/B)")
    ap.add_argument("--commit-behind-replay", action="store_true",
                    help="commit + next step's scalars in ONE eager launch behind each replay (round 3's form; default now: "
                         "last node of the graph, scalars from a device table) (A/B)")
    ap.add_argument("--dynamic-loss-scale", action="store_true",
                    help="(the default since round 4) GradScaler dynamics: init 65536, x2 / 2000 clean steps, x0.5 on overflow "
                         "-- the reference's mixed_precision=True, /root/reference/nerf_vo/mapping/nerfstudio.py:59")
    ap.add_argument("--static-loss-scale", action="store_true",
                    help="A/B: tcnn's static loss scale 128 (rounds 1-3's headline regime: most fp16 proposal-loss gradients "
                         "underflow to exactly zero there and the grid backward skips them)")
    ap.add_argument("--ngp-steps", type=int, default=200,
                    help="timed steps of the occupancy-grid back-end (tools/ngp_bench.py) reported as the ngp section; 0 = skip")
    ap.add_argument("--render-frames", type=int, default=5,
                    help="timed full-image renders per resolution (1200x680 and the training resolution); 0 = skip")
    ap.add_argument("--pmc-traffic", choices=("live", "committed", "off"), default="live",
                    help="roofline.traffic: live = two child runs of this command under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                         "(about a minute; falls back to the committed summary and says so), committed = the latest "
                         "profiles/*_pmc_fetch_write_per_kernel.json, off = null")
    ap.add_argument("--mapping-loop", choices=("on", "off"), default="on",
                    help="the reference's whole mapping run (BASELINE configs[1] 'full mapping loop'): 8192 iterations through the "
                         "MappingModule / Nerfstudio mirrors with the ingest cadence, default kernels, fixed exact poses; wall seconds, "
                         "timed windows as the field trains, kernel table of the trained-field step (tools/mapping_loop.py; ~8 s)")
    ap.add_argument("--pipeline-single-gpu", action="store_true",
                    help="run the next step's sampling prefix beside the fields Adam inside this step's graph (A/B; measured neutral)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher -- it never touches a GPU
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("NVO_BENCH_DRY") == "1":
        raise SystemExit(dry_run(args))
    default_args = ap.parse_args([])  # (what the committed PMC summaries were collected on)
    for a_ in (args, default_args):
        wl = WORKLOADS[a_.workload]
        a_.keyframes = a_.keyframes or wl["keyframes"]
        a_.height = a_.height or wl["height"]
        a_.width = a_.width or wl["width"]
        a_.mlp_dtype = a_.mlp_dtype or wl["mlp_dtype"]
    wl = WORKLOADS[args.workload]
    use_normals = wl["normals"]
    if args.no_cpu_baseline:
        args.cpu_baseline = "off"

    # stdout must carry exactly ONE JSON line: native libraries (RCCL prints a version banner on
    # communicator creation) write to fd 1 directly, so fd 1 is pointed at stderr for the whole run and
    # the result is written to a private duplicate of the original stdout at the end.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE {world} != --gpus {args.gpus} (launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}, or plainly as `python bench.py --gpus {args.gpus}`)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:  # launched by torch.distributed.run (any N)
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="nccl", device_id=device)
        dist = dist_mod

    import __graft_entry__ as entry

    if rank == 0:
        entry.build()
    if dist is not None:
        dist.barrier()
    from nerf_vo_amd import _lib
    from nerf_vo_amd.engine import EngineConfig, NerfactoEngine
    from nerf_vo_amd.mapping.dataset import DynamicDataManager, DynamicDataManagerConfig, opencv_to_opengl
    from nerf_vo_amd.parallel import GradientAllReduce
    from nerf_vo_amd.synthetic import make_sequence

    torch.manual_seed(42 + rank)
    # ---- resident synthetic keyframe buffer (ingested through the same path the mapper uses)
    dm = DynamicDataManagerConfig(train_num_rays_per_batch=args.rays, num_frames=args.keyframes,
                                  frame_height=args.height, frame_width=args.width, use_normals=use_normals).setup(
        device=device, world_size=world, local_rank=rank)
    chunk = 24
    # (render the sequence once, ingest in tracker-sized chunks)
    seq = make_sequence(args.keyframes, args.height, args.width, device=device)
    for lo in range(0, args.keyframes, chunk):
        hi = min(args.keyframes, lo + chunk)
        dm.train_dataset.update({
            "keyframe_indices": torch.arange(lo, hi),
            "camera_intrinsics": seq["camera_intrinsics"][lo:hi],
            "camera_extrinsics": opencv_to_opengl(seq["camera_extrinsics"][lo:hi]),
            "frames_color": seq["frames_color"][lo:hi], "frames_depth": seq["frames_depth"][lo:hi],
            **({"frames_normal": seq["frames_normal"][lo:hi]} if use_normals else {})})
    del seq
    ds = dm.train_dataset
    assert ds.num_active_frames == args.keyframes

    cfg = EngineConfig(num_images=args.keyframes, num_rays=args.rays, optimize_poses=args.optimize_poses,
                       mlp_dtype=args.mlp_dtype, expect_normals=use_normals)
    if args.pipeline_single_gpu:
        cfg.pipeline_single_gpu = True
    cfg.dynamic_loss_scale = not args.static_loss_scale
    if args.commit_in_graph:
        cfg.commit_behind_replay = cfg.commit_from_table = False
    if args.commit_behind_replay:
        cfg.commit_from_table = False
    if args.no_fuse_grid_adam:
        cfg.fuse_grid_adam = False
    if os.environ.get("NVO_DW_REPLICAS") is not None:  # A/B
        cfg.dw_replicas = int(os.environ["NVO_DW_REPLICAS"])
    if args.no_pose_overlap:
        cfg.overlap_pose_backward = False
    if args.no_overlap:
        cfg.overlap_proposal_backward = False
        cfg.overlap_pose_backward = False
    if args.grid_bwd_mode is not None:
        cfg.grid_bwd_mode = args.grid_bwd_mode[0] if len(args.grid_bwd_mode) == 1 else tuple(args.grid_bwd_mode)
    bwd_modes = cfg.grid_bwd_mode if isinstance(cfg.grid_bwd_mode, (tuple, list)) else (cfg.grid_bwd_mode,) * 3
    engine = NerfactoEngine(cfg, device, world_size=world, rank=rank)
    compress = None
    if dist is not None:
        # bf16 is the faithful 2-byte wire format (nerf_vo_amd/parallel.py); should this RCCL build refuse the dtype,
        # every rank sees the same error here and the exchange falls back to fp16 instead of ending the run
        compress = os.environ.get("NVO_GRAD_COMPRESS", "bf16")
        if compress == "bf16":
            try:
                probe = torch.ones(64, dtype=torch.bfloat16, device=device)
                dist.all_reduce(probe)
                torch.cuda.synchronize(device)
                if float(probe[0]) != float(world):
                    raise RuntimeError("bf16 all-reduce returned a wrong sum")
            except Exception as exc:  # noqa: BLE001
                sys.stderr.write(f"[bench] bf16 all-reduce unavailable ({exc}); using fp16 compression\n")
                compress = "fp16"
        if compress == "none":
            compress = None
    # sharded optimiser (reduce-scatter -> Adam on this rank's slice -> all-gather of the working copy): the default
    # whenever the exchange is compressed; NVO_SHARD_OPT=0 = all-reduce + replicated optimiser
    shard_opt = compress is not None and os.environ.get("NVO_SHARD_OPT", "1") != "0"
    reducer = GradientAllReduce(dist, compress=compress, shard_optimizer=shard_opt) if dist is not None else None
    if dist is not None:  # identical initial parameters on every rank
        dist.broadcast(engine.params, src=0)
        engine.sync_half()
    intr = ds.camera_intrinsics
    c2w = ds.camera_extrinsics[:, :3, :4].contiguous()

    use_graph = not args.no_graph

    def step():
        nonlocal use_graph
        if use_graph:
            engine.train_step_graphed(ds, all_reduce=reducer)
        else:
            ray_indices, _ = dm.next_train(engine.step)
            engine.train_step(ray_indices, intr, c2w, ds.frames_color, ds.frames_depth, all_reduce=reducer,
                              normals=ds.world_normals01() if use_normals else None)

    def fence():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    samples_per_step = args.rays * cfg.num_nerf_samples * world
    value = samples_per_step / (elapsed / args.steps)
    losses = engine.loss_dict()

    # ---- roofline: same K steps again with per-launch HIP events on the launch stream
    roofline = None
    live_summary_path = None
    kernel_table = []
    lib = _lib.lib()
    prof_steps = 0 if args.no_kernel_table else min(args.steps, 50)
    if rank == 0 and prof_steps:
        lib.nvo_profile_enable(1)
    fence()
    tp0 = time.perf_counter()
    use_graph_saved, use_graph = use_graph, False  # launchers (and their event hooks) only run eagerly
    # per-kernel durations are taken with every kernel alone on the GPU: the timed region above overlaps the
    # proposal backward with the main-field backward on a second stream, which would inflate both here
    overlap_saved, cfg.overlap_proposal_backward = cfg.overlap_proposal_backward, False
    overlap_pose_saved, cfg.overlap_pose_backward = cfg.overlap_pose_backward, False
    # Eager launches with an event pair each are host-bound (the host needs longer per step than the GPU does), so a
    # multi-kernel scope such as grid_bwd_stream would time the idle gaps between its kernels as well.  A spin of
    # about one step's host time at the head of every profiled step lets the host run ahead: the scopes then see
    # their kernels back to back, which is what rocprofv3's per-kernel durations add up to.
    spin_cycles = 0
    if hasattr(torch.cuda, "_sleep"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1_000_000)
        torch.cuda.synchronize(device)
        e0.record()
        torch.cuda._sleep(1_000_000)
        e1.record()
        torch.cuda.synchronize(device)
        ms_per_mcycle = max(e0.elapsed_time(e1), 1e-3)
        spin_cycles = int(1_000_000 * min(2.0, max(0.3, 1.5 * ms_per_step)) / ms_per_mcycle)
    for _ in range(prof_steps):  # every rank runs them (the all-reduce is collective)
        if spin_cycles:
            torch.cuda._sleep(spin_cycles)
        step()
    use_graph = use_graph_saved
    cfg.overlap_proposal_backward = overlap_saved
    cfg.overlap_pose_backward = overlap_pose_saved
    fence()
    ms_prof = (time.perf_counter() - tp0) / max(prof_steps, 1) * 1e3
    if rank == 0 and prof_steps:
        need = lib.nvo_profile_summary(None, 0)
        buf = C.create_string_buffer(int(need) + 16)
        lib.nvo_profile_summary(buf, len(buf))
        lib.nvo_profile_enable(0)
        for line in buf.value.decode().strip().splitlines():
            name, cnt, total = line.rsplit(",", 2)
            kernel_table.append((name, int(cnt), float(total)))
        kernel_table.sort(key=lambda r: -r[2])
        tot = sum(r[2] for r in kernel_table)
        sys.stderr.write(f"[bench] per-kernel device time over {prof_steps} profiled steps "
                         f"({ms_prof:.3f} ms/step wall, {tot / prof_steps:.3f} ms/step in kernels):\n")
        for name, cnt, total in kernel_table:
            sys.stderr.write(f"[bench]   {name:28s} launches {cnt:5d}  avg {total / cnt * 1e3:9.1f} us  "
                             f"{100 * total / tot:5.1f} %\n")
        for name, cnt, total in kernel_table:  # dominant kernel with a byte model
            fused_plan = engine._fused_adam_plan() if world == 1 else None
            fused_n = (fused_plan[1] - fused_plan[0]) if fused_plan else 0
            b = algorithmic_bytes(name, cfg, args.rays, fused_n)
            if b is None:
                continue
            avg_s = total / cnt * 1e-3
            achieved = b / avg_s / 1e9
            live_path = live_err = None
            if args.pmc_traffic == "live" and world == 1:
                live_path, live_err = live_pmc_summary(workload_argv(args))
                live_summary_path = live_path
                if live_err:
                    sys.stderr.write(f"[bench] live PMC passes failed ({live_err}); using the committed summary\n")
            # a COMMITTED summary was collected on the default command line: it says nothing about another configuration
            default_cfg = workload_argv(args) == workload_argv(default_args)
            if args.pmc_traffic == "off" or (live_path is None and not default_cfg):
                traffic, traffic_src, traffic_raw = None, ("no PMC summary for this configuration" if args.pmc_traffic != "off" else None), None
            else:
                traffic, traffic_src, traffic_raw = pmc_traffic(name, live_path)
            if live_err and traffic_src:
                traffic_src += f" (live passes failed: {live_err[:80]})"
            roofline = {"kernel": name, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "traffic_source": traffic_src,
                        # uncorrected counters (FETCH_SIZE counts half of the lines read on gfx950: calibrated for coalesced
                        # streams and for the record pass's unaligned 384-byte runs, profiles/r3_pmc_probe_run_gather.txt)
                        "traffic_raw": traffic_raw,
                        "avg_launch_us": round(avg_s * 1e6, 2), "algorithmic_bytes_per_launch": int(b),
                        "bytes_model": "SURVEY.md 8d per-unit bytes x units per launch (DESIGN.md section 3)" + (
                            f" + 28 B x {fused_n} parameters stepped inside the pass" if (fused_n and name == "grid_bwd_stream[L16]") else ""),
                        "share_of_step_kernel_time": round(total / tot, 4)}
            if name.startswith("grid_bwd"):
                # informational: what the kernel really read-modify-writes (fp32 gradient pairs, 2x the 8d figure)
                lv = 16 if name.endswith("[L16]") else 5
                extra = 28 * fused_n if name == "grid_bwd_stream[L16]" else 0  # (the Adam bytes are fp32 already)
                units = (b - extra) / (lv * 8 * 4 * 2 + 12 + lv * 4)
                b32 = units * (lv * 8 * 8 * 2 + 12 + lv * 4) + extra
                roofline["fp32_rmw_variant"] = {"algorithmic_bytes_per_launch": int(b32),
                                                "achieved": round(b32 / avg_s / 1e9, 2),
                                                "frac": round(b32 / avg_s / 1e9 / HBM_PEAK_GBS, 5)}
            break
    roofline_gather = gather_roofline(kernel_table, cfg, args.rays, live_summary_path) if rank == 0 else None
    if live_summary_path and os.path.basename(os.path.dirname(live_summary_path)).startswith("nvo_pmc_"):
        shutil.rmtree(os.path.dirname(live_summary_path), ignore_errors=True)  # (the passes' traces: 25 MB per run; a copy of the summary is in gpurun_out/)
    # MFMA utilisation of the fused-MLP kernels alone: FLOP model 2*N*(I*W + (H-1)*W*W + W*O)
    # for a forward; a backward = input gradient + weight gradient + (these networks store no hidden activations)
    # the recomputed forward = 3x.  N = main-field samples per launch; dense fp16 MFMA peak 2.5 PFLOP/s.
    mlp_mfma = []
    if rank == 0:
        import re as _re

        n_main = args.rays * cfg.num_nerf_samples
        for name, cnt, total in kernel_table:
            m = _re.fullmatch(r"mlp_(fwd|bwd)\[(\d+)-(\d+)x(\d+)-(\d+)\]", name)
            if not m or int(m.group(3)) != 64:  # the 16-wide proposal networks run on two batch sizes: not modelled
                continue
            i_, w_, h_, o_ = (int(m.group(k)) for k in (2, 3, 4, 5))
            flop = 2.0 * n_main * (i_ * w_ + (h_ - 1) * w_ * w_ + w_ * o_) * (1 if m.group(1) == "fwd" else 3)
            tflops = flop / (total / cnt * 1e-3) / 1e12
            mlp_mfma.append({"kernel": name, "avg_launch_us": round(total / cnt * 1e3, 2), "achieved": round(tflops, 1),
                             "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tflops / 2500.0, 4)})
    if dist is not None:
        dist.barrier()

    # ---- informational: the same K-step measurement late in the schedule.  `value` above is taken over steps
    # [warmup, warmup + K) of a fresh run, where nerfacto refreshes its proposal networks every 2nd step; past the
    # 5000-step proposal warm-up they are refreshed every 6th step (the reference trains 8192 steps per sequence).
    late = None
    if args.late_steps > 0:
        engine.step = 6000
        for _ in range(12):
            step()
        fence()
        tl0 = time.perf_counter()
        for _ in range(args.late_steps):
            step()
        fence()
        t_late = time.perf_counter() - tl0
        if dist is not None:
            t = torch.tensor([t_late], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_late = float(t.item())
        late = {"first_step": 6012, "steps": args.late_steps, "ms_per_step": t_late / args.late_steps * 1e3,
                "value": samples_per_step / (t_late / args.late_steps),
                "note": "proposal networks refreshed every 6th step (schedule past the 5000-step warm-up)"}

    # ---- inference render (the second caller north_star names: NerfstudioRenderer.render_frame,
    # /root/reference/evaluation/nerf_renderer.py:132-168): full frames at the dataset's native resolution (1200x680 for
    # Replica: evaluation_frame_* keys absent, /root/reference/run.py:57-66) and at the training resolution, chunks of
    # 32 768 rays (nerfstudio's eval_num_rays_per_chunk), one hipGraph per image, inputs already on the device
    render = None
    if rank == 0 and world == 1 and args.render_frames > 0:
        from nerf_vo_amd.mapping.cameras import Cameras
        from nerf_vo_amd.synthetic import replica_intrinsics

        chunk = 1 << 15
        per_ray = cfg.num_nerf_samples * (16 * 8 * 4 + 12 + 16 * 4) + sum(cfg.num_proposal_samples) * (5 * 8 * 4 + 12 + 5 * 4)
        render = {"chunk_rays": chunk, "launch": "one hipGraph per image (all chunks; mean appearance embedding once)",
                  "outputs": "rgb, median depth, expected depth, accumulation (normals on first access only)",
                  "bytes_model": "SURVEY.md 8d forward bytes: 588 B per main-field sample + 192 B per proposal sample",
                  "algorithmic_bytes_per_ray": per_ray, "frames": []}
        pose = ds.camera_extrinsics[:1, :3, :4].clone()
        for w_, h_ in ((1200, 680), (args.width, args.height)):
            fx, fy, cx, cy = replica_intrinsics(h_, w_)
            cams = Cameras(camera_to_worlds=pose, fx=fx, fy=fy, cx=cx, cy=cy, width=w_, height=h_).to(device)
            n_rays = w_ * h_

            def frame():
                b = cams.generate_rays(camera_indices=0, keep_shape=True)
                return engine.render_image(b.origins.reshape(-1, 3), b.directions.reshape(-1, 3),
                                           b.metadata["directions_norm"].reshape(-1), chunk=chunk)

            frame()  # (captures the graph of this image shape)
            torch.cuda.synchronize(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tw0 = time.perf_counter()
            e0.record()
            for _ in range(args.render_frames):
                out_img = frame()
            e1.record()
            torch.cuda.synchronize(device)
            wall = (time.perf_counter() - tw0) / args.render_frames
            ms = e0.elapsed_time(e1) / args.render_frames
            assert bool(torch.isfinite(out_img["rgb"]).all())
            render["frames"].append({
                "resolution": [w_, h_], "rays": n_rays, "chunks": (n_rays + chunk - 1) // chunk,
                "ms_per_frame": round(ms, 3), "ms_per_frame_wall": round(wall * 1e3, 3),
                "rays_per_sec": n_rays / (ms * 1e-3),
                "field_evals_per_sec": n_rays * (cfg.num_nerf_samples + sum(cfg.num_proposal_samples)) / (ms * 1e-3),
                # The frame's algorithmic bytes are table GATHERS (588 B of 600 per main-field sample, 160 of 192 per proposal
                # sample), and the tables -- 24.4 MB main, 1.5 + 1.7 MB proposal, fp16 -- stay in the 32 MiB of L2 (k_grid_fwd
                # pins a level to one XCD; the PMC counters of the training step show 8.7-12.1 MB fetched per grid_fwd[L5] launch
                # against 201 MB of algorithmic gathers).  The bound that applies is the L2 gather rate, not HBM.
                "roofline": {"bound": "l2", "achieved": round(n_rays * per_ray / (ms * 1e-3) / 1e9, 1), "peak": L2_PEAK_GBS,
                             "unit": "GB/s", "frac": round(n_rays * per_ray / (ms * 1e-3) / 1e9 / L2_PEAK_GBS, 4),
                             "note": "algorithmic gather bytes over the L2 aggregate rate; HBM traffic is a small fraction of "
                                     "these bytes (tables L2-resident), so no HBM fraction is quoted"}})
        # per-kernel table of one 1200x680 frame, eagerly with HIP events on the launch stream
        if not args.no_kernel_table:
            fx, fy, cx, cy = replica_intrinsics(680, 1200)
            cams = Cameras(camera_to_worlds=pose, fx=fx, fy=fy, cx=cx, cy=cy, width=1200, height=680).to(device)
            b = cams.generate_rays(camera_indices=0, keep_shape=True)
            torch.cuda.synchronize(device)
            lib.nvo_profile_enable(1)
            engine.render_image(b.origins.reshape(-1, 3), b.directions.reshape(-1, 3), b.metadata["directions_norm"].reshape(-1),
                                chunk=chunk, use_graph=False)
            torch.cuda.synchronize(device)
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
            sys.stderr.write(f"[bench] render 1200x680, eager: {tot:.2f} ms in kernels per frame\n")
            table = []
            for name, cnt, total in rows:
                sys.stderr.write(f"[bench]   {name:28s} launches {cnt:5d}  avg {total / cnt * 1e3:9.1f} us  {100 * total / tot:5.1f} %\n")
                row = {"kernel": name, "launches": cnt, "avg_launch_us": round(total / cnt * 1e3, 2), "share": round(total / tot, 4)}
                bts = algorithmic_bytes(name, cfg, chunk)
                if bts is not None and name.startswith("grid_fwd"):
                    # (launches of the ragged last chunk are smaller: the frame's samples over the frame's time)
                    # (algorithmic_bytes prices ONE launch -- for the two proposal grids their average)
                    frame_bytes = bts / chunk * 1200 * 680 * (2 if "L5" in name else 1)
                    # (cache-resident tables: priced against the L2 gather rate, see the frame's roofline above)
                    row["roofline"] = {"bound": "l2", "achieved": round(frame_bytes / (total * 1e-3) / 1e9, 1),
                                       "peak": L2_PEAK_GBS, "unit": "GB/s"}
                    row["roofline"]["frac"] = round(row["roofline"]["achieved"] / L2_PEAK_GBS, 4)
                table.append(row)
            render["kernel_table_1200x680"] = table

    # ---- the occupancy-grid ("instant-ngp") back-end of the same hot path (SURVEY.md section 8 row a13;
    # /root/reference/nerf_vo/mapping/instant_ngp.py:104-105 -> Testbed.frame()): step time through the pyngp facade
    ngp = None
    if rank == 0 and world == 1 and args.ngp_steps > 0:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import ngp_bench

        # on the reference's own configuration of this back-end (configs/nerf_slam_replica.yaml:14-19: 192 keyframes 360x640,
        # compute_covariances: True -> a non-constant per-pixel depth variance, so the variance gather runs)
        ngp = ngp_bench.run(argparse.Namespace(steps=args.ngp_steps, warmup=300, keyframes=192, height=360, width=640,
                                               cov="varying", extrinsics=1, render_frames=3,
                                               profile=not args.no_kernel_table), quiet=True)

    # ---- CPU baseline: the torch-CPU oracle of the same step on a bounded sample (rank 0, N=1)
    cpu_baseline = None
    if rank == 0 and world == 1 and args.cpu_baseline != "off":
        from oracle.bench_cpu import time_cpu_step

        if args.cpu_baseline == "configs0":  # BASELINE configs[0] as written
            # median of >= 20 steps after 3 warm-ups (2.7 s per step on the GPU box's host: ~1 min)
            cpu_baseline = time_cpu_step(num_rays=4096, num_images=1, steps=20, warmup=3, keyframe=(480, 640),
                                         max_threads=None)
        else:  # explicitly labelled fallback: a 256-ray sample of the same step
            cpu_baseline = time_cpu_step(num_rays=args.cpu_baseline_rays, num_images=8)

    # ---- the reference's whole mapping run (BASELINE configs[1]: "full mapping loop"): MappingModule.step -> Nerfstudio(update |
    # train), 8192 iterations with the ingest cadence, default (non-deterministic) kernels, fixed exact poses
    mapping_loop = None
    if rank == 0 and world == 1 and args.mapping_loop == "on" and args.workload in ("replica", "replica360"):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import mapping_loop as mapping_loop_mod

        mapping_loop = mapping_loop_mod.run(keyframes=args.keyframes, height=args.height, width=args.width, iterations=8192,
                                            profile_steps=0 if args.no_kernel_table else 60,
                                            camera_optimizer_mode="SE3" if args.optimize_poses else "off")

    # ---- render PSNR (rank 0, N=1): outside the timed region, separate end-to-end mapping runs scored by the reference's
    # PUBLISHED protocol (tools/eval_protocol.py: frame-0 pose alignment + median depth scale, evaluation frames at
    # ground-truth poses carried into the model's world, JPEG / 16-bit PNG files, the reference's metrics --
    # /root/reference/evaluation/renderer.py:79-124,239-298, evaluator.py:88-146)
    render_psnr = None
    if rank == 0 and world == 1 and args.psnr != "off" and args.workload == "replica":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from eval_protocol import run as run_protocol

        kw = (dict(keyframes=48, height=240, width=320, iterations=1500, eval_frames=6) if args.psnr == "small" else
              dict(keyframes=192, height=480, width=640, iterations=8192, eval_frames=6))
        # Every run uses the DETERMINISTIC mode with a fixed seed (bitwise reproducible training: DESIGN.md section 3.7), so
        # the figures are the same on every box and every run.  The training THROUGHPUT is the timed region above, not
        # these runs (deterministic mode is ~2x slower).
        def psnr_run(mode, noise=None, seed=42):
            res = run_protocol(quiet=True, deterministic=True, seed=seed, camera_optimizer_mode=mode, pose_noise=noise,
                               keyframe_views=False, **kw)
            ef = res["evaluation_frames"]
            return {"psnr_reference_uint8wrap_db": round(ef["psnr"], 3), "psnr_float_mse_db": round(ef["psnr_float_mse"], 3),
                    "mssim": round(ef["mssim"], 4), "depth_l1": round(ef["absolute_difference"], 4),
                    "depth_delta1": round(ef["delta1"], 4), "scale_pred2gt": round(res["scale_pred2gt"], 5),
                    "pose_rotation_error_rad": round(res["pose_error_after_frame0_alignment"]["rotation_mean_rad"], 6),
                    "ingested_pose_rotation_error_rad": round(res["pose_error_of_ingested_poses"]["rotation_mean_rad"], 6),
                    # the same pose error split into the cameras' common rigid motion (a gauge the field absorbs; the
                    # protocol's alignment pins it on frame 0 alone) and what is left per camera
                    "pose_error_gauge_split": {k: round(v, 6) for k, v in res["pose_error_gauge_split"].items()},
                    "loss_scale_end": res["loss_scale_end"], "seed": seed}

        # headline = fixed exact poses, MEDIAN run (by float-MSE PSNR) of three fixed seeds, every run listed: one of 29
        # otherwise identical 8192-step runs measured in round 4 (deterministic, seed 42) ends in an optimisation blow-up --
        # density pre-activations beyond 15, where trunc_exp's backward multiplies by e^15 -- that the step count does not
        # recover from (12 dB); a single fixed seed would either hide that or report it as the typical quality
        seeds = (42, 43, 44) if args.psnr == "replica" else (42,)
        fixed = sorted((psnr_run("off", seed=sd) for sd in seeds), key=lambda r: r["psnr_float_mse_db"])
        render_psnr = {**fixed[len(fixed) // 2], "fixed_pose_runs": fixed, "held_out_views": kw["eval_frames"],
                       "protocol": "reference evaluation protocol (frame-0 alignment, median depth scale, JPEG/PNG files, "
                                   "uint8-wrapping PSNR + conventional float-MSE PSNR of the same files)",
                       "config": f'{kw["keyframes"]} keyframes {kw["width"]}x{kw["height"]} (every 2nd dataset frame), '
                                 f'{kw["iterations"]} iterations through the Nerfstudio mapper interface (incremental keyframe '
                                 "ingest), fixed exact poses (BASELINE configs[1]), synthetic textured room; deterministic mode, "
                                 "GradScaler loss scale; median of the listed seeds",
                       # the mapper exactly as the reference configures it (SE3 refinement on), exact poses
                       "with_se3_pose_refinement": psnr_run("SE3"),
                       # BASELINE configs[2]: tracker-like pose errors (sigma 5e-3 rad / 5e-3 units, frame 0 exact) with
                       # and without the refinement -- what the camera optimiser is there for
                       "noisy_poses_fixed": psnr_run("off", (5e-3, 5e-3)),
                       "noisy_poses_se3_refinement": psnr_run("SE3", (5e-3, 5e-3))}

    if rank == 0:
        captured = any(e.get("captured_collectives") for e in engine._graphs.values())
        if not use_graph:
            launch_desc = "eager"
        elif dist is None and any(e.get("pipelined") for e in engine._graphs.values()):
            launch_desc = ("hipGraph replay: ONE graph per step = body -> Adam(proposal, poses) -> [Adam(fields) || sampling "
                           "prefix of the next step] (variants: with / without proposal update, value-only proposal losses)")
        elif dist is None:
            launch_desc = "hipGraph replay: ONE graph per step (variants: with / without proposal update, value-only proposal losses)"
        elif captured:
            launch_desc = ("hipGraph replay: ONE graph per step with the RCCL collectives captured inside "
                           "(body -> exchange -> next sampling prefix)")
        else:
            launch_desc = "hipGraph replay: compute segments (body / opt_a / head0 / opt_b / head1) around eager collectives"
        if dist is None:
            parallelism = "single GPU (no process group)"
        else:
            wire = compress or "fp32"
            parallelism = (f"rays sharded x{world} (torch.distributed, backend nccl = RCCL), per step: all-reduce of the "
                           f"proposal / pose gradients + " +
                           (f"reduce-scatter ({wire}) of the fields gradient -> Adam on the rank's 1/{world} slice -> "
                            f"all-gather of the 16-bit working copy" if shard_opt else
                            f"all-reduce ({wire}) of the fields gradient, replicated Adam"))
        out = {
            "metric": "training ray-samples/sec", "value": value, "unit": "ray-samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.mlp_dtype, "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]: Replica-shaped full mapping step, depth-nerfacto (proposal "
                                    "sampling 256/96/48), SE3 pose-gradient backprop enabled"
                                    if (args.optimize_poses and args.workload == "replica") else
                                    wl["name"] + (", SE3 pose-gradient backprop enabled" if args.optimize_poses else "")),
                       "normal_supervision": use_normals,
                       # the reference's regime (mixed_precision=True -> torch GradScaler around tcnn's 16-bit networks)
                       "loss_scale": ("GradScaler 65536 dynamic" if cfg.dynamic_loss_scale else
                                      "static 128 (tcnn default; A/B only)"),
                       # orientation / predicted-normal losses have multiplier 0 in the reference's configuration
                       # (nerfstudio.py:74-75): exactly zero loss and gradient, their heads are not evaluated
                       "skipped_zero_weight_heads": True,
                       "rays_per_gpu": args.rays, "samples_per_ray": cfg.num_nerf_samples,
                       "proposal_samples": list(cfg.num_proposal_samples), "keyframes": args.keyframes,
                       "resolution": [args.width, args.height], "sampler": "proposal-network (nerfacto)",
                       "grid_bwd": [{0: "atomic", 1: "lds", 2: "binned", 3: "stream"}[int(m)] for m in bwd_modes],
                       "launch": launch_desc, "parallelism": parallelism,
                       # which form of the optimiser ran (the Adam step of the main grid's hashed levels inside the grid
                       # backward needs the step's overflow verdict before the exchange: single GPU only)
                       "optimizer": ("fused grid Adam: on (hashed levels stepped inside the grid backward; one launch for the rest)"
                                     if (world == 1 and engine._fused_adam_plan()) else
                                     "fused grid Adam: off" + (" under a process group (verdict travels with the exchange); " +
                                                               ("sharded Adam on 1/%d of the fields group" % world if shard_opt
                                                                else "replicated Adam") if dist is not None else ""))},
            "rays_per_sec": args.rays * world / (elapsed / args.steps),
            # main + both proposal levels: every field evaluation a ray costs
            "field_evals_per_sec": args.rays * world * (cfg.num_nerf_samples + sum(cfg.num_proposal_samples))
            / (elapsed / args.steps),
            "mlp_mfma": mlp_mfma,
            "roofline_gather": roofline_gather,
            "final_losses": losses,
            "late_schedule": late,
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "mapping_loop": mapping_loop,
            "render_psnr": render_psnr,
            "render": render,
            "ngp": ngp,
        }
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
