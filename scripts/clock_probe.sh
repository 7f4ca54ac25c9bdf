#!/bin/bash
# Samples the GPU's shader clock and socket power (rocm-smi) while a per-shape conv benchmark loops: DVFS evidence for DESIGN.md.
# usage (on the GPU box): bash scripts/clock_probe.sh [--zeros]
cd "$(dirname "$0")/.."
python scripts/bench_conv.py --what fwd --only "16^2 512" --iters 60000 $1 > gpurun_out/clock_probe_bench.log 2>&1 &
pid=$!
sleep 12
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.5
done
kill $pid 2>/dev/null
wait $pid 2>/dev/null
echo idle:
sleep 2
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ';'
echo
