#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel (name, grid) the mean of every counter per launch.
Usage: python tools/pmc_kernel.py <dir> [kernel-substring]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                m = re.search(r"(k_\w+)", row["Kernel_Name"])
                if not m or (len(sys.argv) > 2 and sys.argv[2] not in m.group(1)):
                    continue
                a = agg[(m.group(1), int(row["Grid_Size"]))][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    for key in sorted(agg):
        print(f"{key[0]} grid={key[1]}")
        for c, (tot, n) in sorted(agg[key].items()):
            print(f"    {c:28s} {tot / n:16.1f}   ({n} launches)")


if __name__ == "__main__":
    main()
