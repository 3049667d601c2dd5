#!/usr/bin/env bash
# LDS counters of the sample kernel over bench.py's batch-256 workload (one --pmc pass, kernel trace only): how busy the LDS pipe is.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_lds; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv \
   -d $O/lds -- python3 $R/bench.py --batch ${1:-256} --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $O/lds.log 2>&1 || exit 1
cd $R
python3 - "$(find $O/lds -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    if "lpcnet_sample" not in k:
        continue
    c = {n: sum(x) / len(x) for n, x in v.items()}
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    print(k)
    print("  per launch:", {n: round(x) for n, x in c.items()})
    print("  kernel cycles %.0f; LDS pipe busy (SQ_LDS_IDX_ACTIVE / (256 CUs x cycles)) = %.3f; of which bank conflicts %.3f, address conflicts %.3f"
          % (cyc, c["SQ_LDS_IDX_ACTIVE"] / (256 * cyc), c["SQ_LDS_BANK_CONFLICT"] / (256 * cyc), c.get("SQ_LDS_ADDR_CONFLICT", 0) / (256 * cyc)))
    print("  LDS instructions per sample and workgroup %.0f; LDS pipe cycles per sample and workgroup %.0f"
          % (c["SQ_INSTS_LDS"] / (256 * 98 * 160), c["SQ_LDS_IDX_ACTIVE"] / (256 * 98 * 160)))
PY
