#!/usr/bin/env bash
# Average LDS / VMEM latency inside the two sample-rate kernels: LEVEL counters / instruction counts (development aid).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_latency; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for M in 4 -1; do
  rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM \
     --output-format csv -d $O/pmc_m$M -- python3 $R/tools/multi_one.py $M 1024 30 2 > $O/m$M.log 2>&1 || { tail -5 $O/m$M.log; }
done
cd $R
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/prof_latency/pmc*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "lpcnet_sample" in k:
            c = {n: sum(x) / len(x) for n, x in v.items()}
            print(f.split("/")[2], k, {n: round(x / 1e6, 1) for n, x in c.items()})
            print("   LDS latency (LEVEL/INSTS, cycles):", round(c["SQ_INST_LEVEL_LDS"] / c["SQ_INSTS_LDS"], 1),
                  " VMEM:", round(c["SQ_INST_LEVEL_VMEM"] / (c["SQ_INSTS_VMEM_RD"] + c.get("SQ_INSTS_VMEM_WR", 0)), 1),
                  " SMEM:", round(c["SQ_INST_LEVEL_SMEM"] / max(c["SQ_INSTS_SMEM"], 1), 1))
PY
