#!/bin/bash
# Sample socket power and clocks (rocm-smi, read-only) while the train step runs: is the step power-limited?
# usage (GPU box): bash scripts/power_probe.sh > gpurun_out/power_probe.log
python bench.py --steps 600 --warmup 3 --no-cpu-baseline --sample-steps 0 > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
BP=$!
sleep 12
for i in $(seq 1 25); do
  rocm-smi --showpower --showclocks --showuse -d 0 2>/dev/null | grep -E "Power|sclk|mclk|busy" | tr '\n' ' '
  echo
  sleep 0.25
done
wait $BP
grep bench gpurun_out/power_bench.err
echo "--- idle"
sleep 2
rocm-smi --showpower --showclocks -d 0 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' '; echo
rocm-smi --showmaxpower -d 0 2>/dev/null | grep -i "power" | head -3
