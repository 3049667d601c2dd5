#!/usr/bin/env bash
# rocprofv3 kernel trace of the gated streaming leg (tools/gated_leg.py): per-kernel statistics and, from the trace's own time stamps,
# how many vocoder launches were in flight at once and whether the tick's kernels really ran beside them.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_gated; mkdir -p $O
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8} DSS_LPCNET_SYNTHETIC=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/gated_leg.py > $O/leg.json 2> $O/trace.log || exit 1
cd $R
python3 - "$(find $O/trace -name '*kernel_trace.csv' | head -1)" "$(find $O/trace -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
samp = sorted((a, b) for a, b, n, _ in ev if "lpcnet_sample" in n)
points = sorted([(a, 1) for a, _ in samp] + [(b, -1) for _, b in samp])
cur = peak = 0
for _, d in points:
    cur += d
    peak = max(peak, cur)
import bisect
starts = [a for a, _ in samp]
def running_at(t):
    return sum(1 for a, b in samp[max(0, bisect.bisect_right(starts, t) - 64): bisect.bisect_right(starts, t)] if a <= t < b)
tick = [(a, b, n) for a, b, n, _ in ev if any(k in n for k in ("hga_fused", "vad_lstm", "speech_gate_kernel"))]
beside = sum(1 for a, b, n in tick if running_at(a) > 0)
queues = sorted({q for _, _, n, q in ev if "lpcnet_sample" in n})
print(f"sample-kernel launches {len(samp)}, at most {peak} in flight at once, on queues {queues}")
print(f"tick kernels (hga_fused / vad_lstm / speech_gate) {len(tick)}, of which {beside} started while a vocoder launch was running "
      f"({100.0 * beside / max(1, len(tick)):.0f} %); their durations when beside one: "
      f"p50 {sorted((b - a) for a, b, n in tick if running_at(a) > 0)[beside // 2] / 1e3 if beside else 0:.1f} us, alone: "
      f"p50 {sorted((b - a) for a, b, n in tick if running_at(a) == 0)[(len(tick) - beside) // 2] / 1e3 if len(tick) > beside else 0:.1f} us")
print(open(sys.argv[2]).read())
PY
