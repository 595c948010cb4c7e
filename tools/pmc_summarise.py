#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: mean counter value per dispatch.
Usage: python tools/pmc_summarise.py <dir-or-csv> [name-filter]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if flt and flt not in name:
                    continue
                m = re.search(r"(k_\w+)(<[^(]*>)?", name)
                short = (m.group(1) + (m.group(2) or ""))[:70] if m else name.split("(")[0][-60:]
                a = agg[short][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    for k, counters in sorted(agg.items()):
        print(k)
        for c, (tot, n) in sorted(counters.items()):
            print(f"    {c:28s} {tot / n:16.1f}  (x{n})")


if __name__ == "__main__":
    main()
