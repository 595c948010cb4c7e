#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of the
bench command into profiles/<round>_pmc_fetch_write_per_kernel.json: HBM-side KB per launch for every
kernel of the library, plus the sums bench.py's `roofline.traffic` reads (`bench_name` rows).

An optional third pass (`--pmc TCP_TCC_READ_REQ_sum`: read requests the vector L1s send to the L2, one per cache line a
wave instruction misses) adds `L2_READ_REQ_per_launch` to every kernel: what a gather kernel is really served by when
its tables are cache-resident (bench.py's `roofline_gather`).

Usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [steps_profiled] [l2_read_req_dir]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def load(path, counter):
    agg = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter and not row["Counter_Name"].startswith(counter):
                    continue
                m = re.search(r"(k_\w+)", row["Kernel_Name"])
                if not m:
                    continue
                key = (m.group(1), int(row["Grid_Size"]))
                agg[key][0] += float(row["Counter_Value"])
                agg[key][1] += 1
    return agg


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out_path = sys.argv[3]
    l2req = load(sys.argv[5], "TCP_TCC_READ_REQ") if len(sys.argv) > 5 and sys.argv[5] else {}
    kernels = []
    # only the kernels of the training step: whatever ran fewer than `min_launches` times in the profiled command is set-up
    # work (torch's GEMMs of the synthetic scene, the eager warm-up's own launches) and stays out of the summary
    min_launches = int(sys.argv[4]) // 3 if len(sys.argv) > 4 else 1
    for key in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(key, [0.0, 0])
        w, nw = write.get(key, [0.0, 0])
        n = max(nf, nw, 1)
        if n < min_launches or key[0].startswith(("k_Ailk", "k_Alik", "k_Ajlk")):
            continue
        kernels.append({"kernel": key[0], "grid_size": key[1], "launches": n,
                        "FETCH_SIZE_KB_per_launch": round(f / max(nf, 1), 1),
                        "WRITE_SIZE_KB_per_launch": round(w / max(nw, 1), 1)})
        if key in l2req and l2req[key][1]:
            kernels[-1]["L2_READ_REQ_per_launch"] = round(l2req[key][0] / l2req[key][1], 1)
    # gfx950: FETCH_SIZE reports exactly HALF the bytes of a coalesced streaming read (MI355X_MICROARCH.md, HBM section:
    # 16 B per lane; calibrated here for 8 B per lane too -- tools/probes/read_bw_probe under --pmc FETCH_SIZE reports
    # 70 688 KB for a 141 312 KB buffer at both widths, profiles/r2_pmc_probe_calibration.txt).  The kernels below read
    # nothing but such streams; every other kernel's reads (4-byte gathers, strided rows) are uncalibrated and stay raw.
    # (k_tl_accumulate_p reads 384-byte runs of 12-byte records at arbitrary record offsets, one dwordx3 per lane.  Round
    # 3 calibrated THAT pattern too -- tools/probes/run_gather_probe under --pmc FETCH_SIZE, profiles/
    # r3_pmc_probe_run_gather.txt: a fully coalesced 12-byte stream counts 101 386 KB for 202 752 KB read (x 0.500), the
    # run gather 67 068 KB for 101 376 KB requested (x 0.662 = 1/2 x 4/3: an unaligned 384-byte run touches four 128-byte
    # lines).  The x2 figure is therefore the traffic the pass really causes, line over-fetch included; the raw value
    # stays beside it in bench.py (roofline.traffic_raw).)
    STREAM_READERS = {"k_tl_accumulate", "k_tl_accumulate_p", "k_st_accumulate", "k_adam_groups", "k_adam",
                      "k_nonfinite_flag_ranges", "k_nonfinite_flag", "k_read"}
    for k in kernels:
        k["FETCH_SIZE_KB_corrected"] = round(k["FETCH_SIZE_KB_per_launch"] * (2.0 if k["kernel"] in STREAM_READERS else 1.0), 1)
    # bench rows: the streamed main-grid backward is several kernels per step (+ the slice-owner launch of its
    # coarse levels = the k_grid_bwd_lds launch with the smallest grid)
    st = [k for k in kernels if k["kernel"].startswith(("k_st_", "k_tl_"))]
    lds = sorted((k for k in kernels if k["kernel"] == "k_grid_bwd_lds"), key=lambda k: k["grid_size"])
    rows = []

    def total(parts, field, steps):
        return round(sum(k[field] * k["launches"] for k in parts) / steps, 1)

    if st:
        parts = st + (lds[:1] if lds else [])
        steps = max(k["launches"] for k in parts if k["kernel"] in ("k_st_scatter", "k_tl_scatter", "k_tl_scatter_p"))
        rows.append({"bench_name": "grid_bwd_stream[L16]", "launches": steps, "parts": [f'{k["kernel"]}/{k["grid_size"]}' for k in parts],
                     "FETCH_SIZE_KB_per_launch": total(parts, "FETCH_SIZE_KB_per_launch", steps),
                     "FETCH_SIZE_KB_corrected": total(parts, "FETCH_SIZE_KB_corrected", steps),
                     "WRITE_SIZE_KB_per_launch": total(parts, "WRITE_SIZE_KB_per_launch", steps)})
        lds = lds[1:]
    if lds:
        n = sum(k["launches"] for k in lds)
        rows.append({"bench_name": "grid_bwd_lds[L5]" if st else "grid_bwd_lds[L16]", "launches": n,
                     "parts": [f'{k["kernel"]}/{k["grid_size"]}' for k in lds],
                     "FETCH_SIZE_KB_per_launch": total(lds, "FETCH_SIZE_KB_per_launch", n),
                     "FETCH_SIZE_KB_corrected": total(lds, "FETCH_SIZE_KB_corrected", n),
                     "WRITE_SIZE_KB_per_launch": total(lds, "WRITE_SIZE_KB_per_launch", n)})
    commit = os.environ.get("NVO_COMMIT", "")
    note = ((f"commit {commit}; " if commit else "") +
            "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) around "
            "`python3 bench.py --steps 20 --warmup 5 --psnr off --cpu-baseline off --late-steps 0 --no-kernel-table --render-frames 0 --ngp-steps 0` (graph-replayed steps; kernels launched fewer than a third as often as the step are left out); raw counter unit KB; "
            "per launch = sum / launches.  FETCH_SIZE_KB_corrected = raw x 2 for the kernels whose reads are pure coalesced "
            "streams (MI355X_MICROARCH.md: gfx950 FETCH_SIZE reports half the bytes of such reads; calibrated with "
            "tools/probes/read_bw_probe at 8 and 16 B per lane), raw for every other kernel (uncalibrated access widths); "
            "k_tl_accumulate_p's 384-byte record runs were calibrated separately (tools/probes/run_gather_probe, "
            "profiles/r3_pmc_probe_run_gather.txt: the counter reports 0.662 of the REQUESTED bytes = half of the 128-byte "
            "lines an unaligned run touches), so its x2 figure is the traffic it causes, over-fetch included.")
    json.dump({"note": note, "kernels": rows + kernels}, open(out_path, "w"), indent=1)
    for r in rows:
        print(r)


if __name__ == "__main__":
    main()
