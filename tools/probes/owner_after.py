"""Durations of k_grid_bwd_lds split by what it scans: the main field's launch (directly behind k_mlp_bwd 32-64x1-16 /
k_live_rows in stream order) against the proposal networks' (rocprofv3 kernel trace directory as argument)."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[-240 * 40:]
main, other = [], []
for k, (s, e, n) in enumerate(rows):
    if "k_grid_bwd_lds" not in n:
        continue
    prev = [rows[j][2] for j in range(max(0, k - 3), k)]
    is_main = any("k_live_rows" in p or "Li32ELi64ELi1ELi16ELi1ELb1ELb0ELb1ELb1" in p for p in prev)
    (main if is_main else other).append((e - s) / 1e3)
for name, v in (("main", main), ("proposal", other)):
    if v:
        v.sort()
        print(f"{name}: {len(v)} launches, mean {sum(v) / len(v):.2f} us, median {v[len(v) // 2]:.2f}, min {v[0]:.2f}, max {v[-1]:.2f}")
