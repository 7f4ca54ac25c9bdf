#!/bin/bash
set -o pipefail
O=gpurun_out
for P in 1 0; do
  DM_CONV_PERSIST=$P python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sample-steps 0 --shape-table $O/shapes_persist$P.txt > $O/b6_$P.json 2> $O/b6_$P.err; echo "bench persist=$P rc $?"
done
echo "== persist=1"; head -30 $O/shapes_persist1.txt
echo "== persist=0"; head -30 $O/shapes_persist0.txt
