#!/bin/bash
# development aid: latency kernel vs pair kernel at several batch sizes (run on the GPU box)
set -e
cd "$(dirname "$0")/.."
for B in 256 512 1024; do
  for M in 1 2; do
    timeout -k 10 300 python tools/quick_time.py $B 100 $M
  done
done
