#!/usr/bin/env python3
"""Condense the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/<tag>_pmc_traffic.json.

    python tools/pmc_summary.py gpurun_out/pmc_fetch_e/fetch_counter_collection.csv \
                                gpurun_out/pmc_write_e/write_counter_collection.csv profiles/r1e_pmc_traffic.json

Per-launch HBM bytes of the dominant kernel, corrected as MI355X_MICROARCH.md's HBM section prescribes (counter unit is
KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, WRITE_SIZE is exact).
"""
import csv
import json
import os
import sys
from collections import defaultdict

DOMINANT = "lpcnet_sample"


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return {k: {"dispatches": len(v), "mean_KB": sum(v) / len(v)} for k, v in acc.items()}


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else "batch 256 x 1-s utterances"
    fetch, write = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    name = next(k for k in fetch if DOMINANT in k)
    f_kb, w_kb = fetch[name]["mean_KB"], write[name]["mean_KB"]
    doc = {
        "git_sha": os.environ.get("DSS_PROFILE_SHA", "unknown"),
        "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 "
                   "--warmup 1 --no-cpu-baseline --no-latency",
        "kernel": name,
        "workload": workload,
        "FETCH_SIZE_KB_per_launch": f_kb,
        "WRITE_SIZE_KB_per_launch": w_kb,
        "hbm_bytes_per_launch_corrected": (2.0 * f_kb + w_kb) * 1024.0,
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced "
                      "reads -> doubled; WRITE_SIZE exact for 16-B/lane stores. Most reads here are 4 B/lane row loads "
                      "(uncalibrated width), so the read side is an upper-bound style estimate.",
        "all_kernels": {k: {"FETCH_SIZE": fetch.get(k), "WRITE_SIZE": write.get(k)} for k in sorted(set(fetch) | set(write))},
    }
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(f"{name}: fetch {f_kb:.0f} KB, write {w_kb:.0f} KB -> {doc['hbm_bytes_per_launch_corrected'] / 1e9:.3f} GB per launch")


if __name__ == "__main__":
    main()
