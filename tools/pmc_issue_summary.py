#!/usr/bin/env python3
"""Condense one rocprofv3 --pmc pass of SQ issue counters over bench.py into profiles/<tag>_pmc_issue.json.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
              --output-format csv -d gpurun_out/pmc_issue -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency
    python tools/pmc_issue_summary.py gpurun_out/pmc_issue/.../counter_collection.csv profiles/r2_b256_pmc_issue.json 256 "<cmd>"

Derived: wave-instructions per sample and workgroup, cycles per sample, and the VALU issue utilisation
= VALU wave-instructions x 2 cycles (a wave64 instruction occupies its SIMD-32 for 2 cycles) / (1024 SIMDs x kernel cycles).
"""
import csv
import json
import os
import sys
from collections import defaultdict

DOMINANT = "lpcnet_sample"


def main():
    path, out, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    cmd = sys.argv[4] if len(sys.argv) > 4 else ""
    utts_per_wg = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    name = next(k for k in acc if DOMINANT in k)
    c = {k: sum(v) / len(v) for k, v in acc[name].items()}
    samples = batch * 98 * 160
    wgs = batch / utts_per_wg
    cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0            # the counter sums the 8 XCDs
    doc = {"git_sha": os.environ.get("DSS_PROFILE_SHA", "unknown"), "command": cmd, "kernel": name, "workload": f"batch {batch} x 1-s utterances (98 synthesised frames each)",
           "samples_per_launch": samples, "counters_per_launch": c,
           "per_sample_per_utterance": {
               "valu_wave_instructions": c.get("SQ_INSTS_VALU", 0) / samples,
               "lds_wave_instructions": c.get("SQ_INSTS_LDS", 0) / samples,
               "salu_wave_instructions": c.get("SQ_INSTS_SALU", 0) / samples,
               "vmem_read_wave_instructions": c.get("SQ_INSTS_VMEM_RD", 0) / samples},
           "derived": {
               "kernel_cycles": cycles,
               "cycles_per_sample_step": cycles / (98 * 160) / max(1.0, batch / utts_per_wg / 256.0) if cycles else None,
               "valu_issue_utilisation": (c.get("SQ_INSTS_VALU", 0) * 2.0 / (1024.0 * cycles)) if cycles else None,
               "workgroups": wgs},
           "note": "valu_issue_utilisation = VALU wave-instructions x 2 cycles / (1024 SIMDs x kernel cycles); kernel cycles = "
                   "GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs).  Packed fp32 instructions count once although they hold "
                   "the SIMD twice as long, so this is a lower bound."}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc["derived"]))


if __name__ == "__main__":
    main()
