#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 240 python -m pytest tests/test_gpu_kernels.py -x -q -s -k "persistent" --timeout 100 > $O/t5.log 2>&1; echo "pytest rc $?" | tee -a $O/t5.log
grep -E "passed|failed|Error|assert" $O/t5.log | tail -12
if grep -q " passed" $O/t5.log && ! grep -q "failed" $O/t5.log; then
for P in 1 0 1 0; do
  DM_CONV_PERSIST=$P python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 0 > $O/bench_r03_persist$P.json 2> $O/bench_r03_persist$P.err; echo "bench persist=$P rc $?"
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_r03_persist$P.json")); r=d["roofline"]
print("persist=$P", d["value"], d["ms_per_step"], "halo frac", r["frac"], "avg us", r["avg_launch_us"], "conv fam ms", r["families"]["ms"].get("conv"))
PY
done
fi
