#!/usr/bin/env bash
# Instruction-cache behaviour of the two sample-rate kernels (development aid).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_icache; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
for M in 4 -1; do
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
     --output-format csv -d $O/pmc_m$M -- python3 $R/tools/multi_one.py $M 1024 30 2 > $O/m$M.log 2>&1 || { tail -5 $O/m$M.log; }
done
cd $R
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" gpurun_out/prof_icache/counters.txt | sort -u | tr '\n' ' '
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/prof_icache/pmc*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "lpcnet_sample" in k:
            print(f.split("/")[2], k, {c: round(sum(x) / len(x) / 1e6, 1) for c, x in v.items()})
PY
