#!/usr/bin/env python3
"""Per-kernel statistics of the TRAINING STEPS in a rocprofv3 --kernel-trace run of bench.py.

The raw `--stats` table of such a run mixes in everything else the process launches: the synthetic-scene generator
(hipBLASLt Cijk_*, at::native elementwise kernels), the graph warm-up, the eager per-kernel pass.  This reads the
kernel-trace CSV instead, keeps the dispatches of THIS library's kernels (k_*) that start after the first replayed step
(= after the last graph capture warm-up: the first k_ray_head whose successor k_ray_head is < 5 ms away marks steady
state) and prints name, calls, total / average / min / max duration and share.

`--last N`: only the last N steps of the trace (e.g. the trained-field end of a full mapping run, tools/mapping_loop.py).

Usage: python tools/rocprof_clean_stats.py <rocprof-output-dir> [--skip-first N] [--last N] [--head k_rays_given] > stats.csv"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip-first") + 1]) if "--skip-first" in sys.argv else 0
    files = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {path}")
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    head = sys.argv[sys.argv.index("--head") + 1] if "--head" in sys.argv else "k_ray_head"  # (first kernel of a step)
    heads = [t0 for t0, _, n in rows if head in n]
    # steady state: the first step start followed by >= 20 further starts all < 5 ms apart
    start = heads[0] if heads else rows[0][0]
    for i in range(len(heads) - 20):
        if all(heads[j + 1] - heads[j] < 5_000_000 for j in range(i, i + 20)):
            start = heads[min(i + skip, len(heads) - 1)]
            break
    if "--last" in sys.argv:
        last = int(sys.argv[sys.argv.index("--last") + 1])
        later = [h for h in heads if h >= start]
        if len(later) > last:
            start = later[-last]
    agg = defaultdict(list)
    for t0, t1, n in rows:
        if t0 < start:
            continue
        m = re.search(r"(k_\w+)(<[^(]*>)?", n)
        if not m:
            continue
        agg[(m.group(1) + (m.group(2) or ""))[:90]].append(t1 - t0)
    total = sum(sum(v) for v in agg.values())
    steps = sum(1 for h in heads if h >= start)
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "calls", "calls_per_step", "total_us", "avg_us", "min_us", "max_us", "percent"])
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), round(len(v) / max(steps, 1), 2), round(sum(v) / 1e3, 1), round(sum(v) / len(v) / 1e3, 2),
                    round(min(v) / 1e3, 2), round(max(v) / 1e3, 2), round(100.0 * sum(v) / total, 2)])
    sys.stderr.write(f"[rocprof_clean_stats] {steps} steps after steady-state start, {len(agg)} kernels, "
                     f"{total / 1e3 / max(steps, 1):.1f} us of kernel time per step\n")


if __name__ == "__main__":
    main()
