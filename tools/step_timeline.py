#!/usr/bin/env python3
"""Timeline of graph-replayed steps from a rocprofv3 --kernel-trace CSV: for the last steps (a step starts at each launch
of the ray head), every kernel with its start / end relative to the step's first kernel and its queue -- what the critical
path of an update step and of a plain step really is.  Usage: step_timeline.py <kernel_trace.csv> [first_step] [n_steps]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
heads = [i for i, r in enumerate(rows) if "k_ray_head" in r["Kernel_Name"]]
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(heads) - 4
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for s in range(first, min(first + n, len(heads) - 1)):
    lo, hi = heads[s], heads[s + 1]
    t0 = int(rows[lo]["Start_Timestamp"])
    print(f"---- step #{s}: {hi - lo} kernels, {(int(rows[hi]['Start_Timestamp']) - t0) / 1e3:.1f} us to the next step's head")
    prev_end = {}
    for r in rows[lo:hi]:
        b, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0][:64]
        q = r["Queue_Id"]
        gap = b - prev_end.get(q, b)
        prev_end[q] = e
        print(f"{name:64s} q{q:>2} start {b:8.1f} end {e:8.1f} dur {e - b:7.1f} gap {gap:6.1f}")
