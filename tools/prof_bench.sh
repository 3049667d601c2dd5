#!/usr/bin/env bash
# Profiles of bench.py for profiles/<tag>_*: kernel-trace stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and
# SQ issue counters.   gpurun -- "bash tools/prof_bench.sh r3 256 1 $(git rev-parse --short HEAD)"   (the box has no .git)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r3}; B=${2:-256}; UPW=${3:-1}; export DSS_PROFILE_SHA=${4:-unknown}      # UPW: utterances per workgroup of the kernel that serves batch B (2 beyond one per CU)
O=$R/gpurun_out/prof_${TAG}_b$B; mkdir -p $O
ARGS="--batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-latency"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py $ARGS > $O/trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py $ARGS > $O/write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv \
   -d $O/issue -- python3 $R/bench.py $ARGS > $O/issue.log 2>&1 || exit 1
cd $R
mkdir -p profiles
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) profiles/${TAG}_b${B}_kernel_stats.csv
python3 tools/pmc_summary.py $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) \
    profiles/${TAG}_b${B}_pmc_traffic.json "batch $B x 1-s utterances"
python3 tools/pmc_issue_summary.py $(find $O/issue -name "*counter_collection.csv" | head -1) profiles/${TAG}_b${B}_pmc_issue.json $B \
    "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py $ARGS" $UPW
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_b${B}_* gpurun_out/profiles_out/
head -5 profiles/${TAG}_b${B}_kernel_stats.csv
