#!/usr/bin/env bash
# tools/scale.sh -- the 1/2/4/8-GPU scaling series of BASELINE.json configs[3] on one node (what the driver's
# SCALE_rNN.json runs): 1024 utterances per GPU, utterance-sharded, one RCCL gather of the int16 PCM on rank 0.
# Prints one JSON line per N.  N=1 is run at 1024 utterances too, so that the series is weak scaling throughout.
set -euo pipefail
cd "$(dirname "$0")/.."
STEPS=${STEPS:-5}; WARMUP=${WARMUP:-2}
for N in ${GPUS:-1 2 4 8}; do
  if [ "$N" = 1 ]; then
    python3 bench.py --gpus 1 --batch 1024 --steps "$STEPS" --warmup "$WARMUP" --no-cpu-baseline --no-latency
  else
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29540 + N)) \
      bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARMUP"
  fi
done
