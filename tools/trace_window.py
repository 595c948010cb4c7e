#!/usr/bin/env python3
"""Prints the launches around the N-th occurrence of a kernel in a rocprofv3 --kernel-trace CSV (start / end relative to
it, queue id) -- to see what really ran beside what.  Usage: trace_window.py <kernel_trace.csv> <substr> [nth] [before] [after]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
hits = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
nth = int(sys.argv[3]) if len(sys.argv) > 3 else len(hits) // 2
before = int(sys.argv[4]) if len(sys.argv) > 4 else 3
after = int(sys.argv[5]) if len(sys.argv) > 5 else 12
i = hits[nth]
t0 = int(rows[i]["Start_Timestamp"])
for r in rows[max(0, i - before):i + after]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:56]
    print(f"{name:56s} q{r['Queue_Id']:>2} start {s:9.1f} end {e:9.1f} dur {e - s:7.1f}")
