#!/usr/bin/env python3
"""Markdown table of the parity margins a GPU test session recorded (tests/conftest.py -> parity_margins.json): per test
function and quantity, the largest |got - ref| in units of the stated bound (1.0 = the tolerance is exhausted), the
relative L1 error and the share of elements outside the bound against the share allowed.
python tools/parity_margins.py gpurun_out/parity_margins.json [--by-case]"""
import json
import re
import sys
from collections import OrderedDict


def main():
    path = sys.argv[1]
    by_case = "--by-case" in sys.argv
    rows = json.load(open(path))
    agg = OrderedDict()
    for r in rows:
        test = r["test"].split("::")[-1]
        if not by_case:
            test = re.sub(r"\[.*\]$", "", test)
        what = re.sub(r"\d+", "#", r["what"]) if not by_case else r["what"]
        key = (test, what)
        a = agg.setdefault(key, {"n": 0, "worst": 0.0, "l1": 0.0, "out": 0.0, "allowed": 0.0, "rtol": set(), "atol": set()})
        a["n"] += 1
        a["worst"] = max(a["worst"], r["worst_error_over_bound"])
        a["l1"] = max(a["l1"], r["rel_l1"])
        a["out"] = max(a["out"], r["outlier_frac"])
        a["allowed"] = max(a["allowed"], r["outlier_frac_allowed"])
        a["rtol"].add(r["rtol"])
        a["atol"].add(r["atol_scale"])
    print("| test | quantity | comparisons | rtol | atol x max|ref| | worst error / bound | worst relative L1 | outside the bound (allowed) |")
    print("|---|---|---|---|---|---|---|---|")
    def rng(s):
        if any(isinstance(v, str) for v in s):  # (rows that state another kind of bound: _assert_close_chain)
            return "; ".join(sorted(str(v) for v in s))
        return f"{min(s):.1e}" if len(s) == 1 else f"{min(s):.1e}..{max(s):.1e}"

    for (test, what), a in agg.items():
        print(f"| `{test}` | {what} | {a['n']} | {rng(a['rtol'])} | {rng(a['atol'])} | {a['worst']:.2f} | {a['l1']:.1e} | "
              f"{a['out']:.1e} ({a['allowed']:.0e}) |")


if __name__ == "__main__":
    main()
