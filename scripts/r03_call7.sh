#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q --timeout 300 > $O/t7full.log 2>&1; echo "full pytest rc $?" | tee -a $O/t7full.log
tail -5 $O/t7full.log
grep -q " passed" $O/t7full.log || exit 1
python bench.py --steps 20 --warmup 5 > $O/bench_r03_v3.json 2> $O/bench_r03_v3.err; echo "bench rc $?"; tail -2 $O/bench_r03_v3.err
python bench.py --steps 20 --warmup 5 --force-dp --no-cpu-baseline --sample-steps 0 > $O/bench_r03_v3_dp.json 2> $O/bench_r03_v3_dp.err; echo "bench dp rc $?"; tail -2 $O/bench_r03_v3_dp.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sample-steps 0 > $O/bench_r03_v3_b.json 2> $O/bench_r03_v3_b.err; echo "bench (again) rc $?"
python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_r03_cfg5.json 2> $O/bench_r03_cfg5.err; echo "bench cfg5 rc $?"; tail -3 $O/bench_r03_cfg5.err
python - <<'PY'
import json
for f in ("bench_r03_v3", "bench_r03_v3_dp", "bench_r03_v3_b", "bench_r03_cfg5"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json")); r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], "frac", r["frac"], "step", r["step"], "launches", r["families"] and r["families"]["launches"], "non-mfma", r["families"] and r["families"]["non_mfma_ms"], "sample", d.get("sample") and d["sample"].get("steps_per_s"))
    except Exception as e:
        print(f, "ERR", e)
PY
