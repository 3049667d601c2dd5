#!/usr/bin/env bash
# rocprofv3 over the HGA kernels (kernel-trace stats, then one SQ counter pass), for both forms.   gpurun -- 'bash tools/prof_hga.sh r3'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r3}
O=$R/gpurun_out/prof_hga_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for P in 0 3; do
  export DSS_HGA_PATH=$P
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$P -- python3 $R/tools/hga_prof_run.py > $O/trace$P.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv \
     -d $O/pmc$P -- python3 $R/tools/hga_prof_run.py > $O/pmc$P.log 2>&1 || exit 1
done
cd $R
mkdir -p gpurun_out/profiles_out
for P in 0 3; do
  cp $(find $O/trace$P -name "*kernel_stats.csv" | head -1) gpurun_out/profiles_out/${TAG}_hga_path${P}_kernel_stats.csv
  cp $(find $O/pmc$P -name "*counter_collection.csv" | head -1) gpurun_out/profiles_out/${TAG}_hga_path${P}_counters.csv
  head -6 gpurun_out/profiles_out/${TAG}_hga_path${P}_kernel_stats.csv
done
