#!/usr/bin/env bash
# SQ counters of the sample-rate kernel, throughput form vs latency form (development aid).
#   gpurun -- 'bash tools/prof_multi.sh [batch] [frames]'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=${1:-1024}; F=${2:-30}
O=$R/gpurun_out/prof_multi; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for M in 4 -1; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU \
     --output-format csv -d $O/pmc1_m$M -- python3 $R/tools/multi_one.py $M $B $F 2 > $O/m$M.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
     --output-format csv -d $O/pmc2_m$M -- python3 $R/tools/multi_one.py $M $B $F 2 >> $O/m$M.log 2>&1 || exit 1
done
cd $R
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/prof_multi/pmc*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "lpcnet_sample" in k:
            print(f.split("/")[2], k, {c: round(sum(x) / len(x) / 1e6, 1) for c, x in v.items()})
PY
