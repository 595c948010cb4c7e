#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and prints, for the last occurrences of two kernels (substring match), their
start/end offsets -- to see whether two launches on different streams really ran side by side.
Usage: trace_overlap.py <kernel_trace.csv> <substr A> <substr B> [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
a, b = sys.argv[2], sys.argv[3]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 6
sel = [r for r in rows if a in r["Kernel_Name"] or b in r["Kernel_Name"]]
sel.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(sel[-2 * n]["Start_Timestamp"])
for r in sel[-2 * n:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    g = r.get("Grid_Size", r.get("Grid_Size_X", "?"))
    print(f"{r['Kernel_Name'][:60]:60s} grid {g:>8} queue {r.get('Queue_Id','?'):>3}  start {s/1e3:9.1f} us  end {e/1e3:9.1f} us  dur {(e-s)/1e3:7.1f}")
