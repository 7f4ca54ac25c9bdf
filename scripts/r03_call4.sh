#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_kernels.py -q -s -k "fused_coordattn or fused_se" > $O/t4.log 2>&1; echo "pytest rc $?" | tee -a $O/t4.log
grep -E "CoordAttn B|passed|failed|Error" $O/t4.log | tail -30
python -m pytest tests -m gpu -x -q > $O/t4b.log 2>&1; echo "full pytest rc $?" | tee -a $O/t4b.log
tail -8 $O/t4b.log
python bench.py --steps 20 --warmup 5 > $O/bench_r03_v2.json 2> $O/bench_r03_v2.err; echo "bench rc $?"
tail -3 $O/bench_r03_v2.err
DM_FUSED_CHAINS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sample-steps 0 > $O/bench_r03_v2_unfused.json 2> $O/bench_r03_v2_unfused.err; echo "bench(unfused) rc $?"
python - <<'PY'
import json
for f in ("gpurun_out/bench_r03_v2.json", "gpurun_out/bench_r03_v2_unfused.json"):
    try:
        d = json.load(open(f))
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], "frac", r["frac"], "step", r["step"], "fam", r["families"] and (r["families"]["launches"], r["families"]["ms"], r["families"]["non_mfma_ms"]))
        print("   hbm", r["hbm_bound"]); print("   sample", d.get("sample"))
    except Exception as e:
        print(f, "ERR", e)
PY
