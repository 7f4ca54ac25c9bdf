#!/bin/bash
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused" --timeout 120 > $O/t12.log 2>&1; echo "pytest rc $?"; tail -3 $O/t12.log
grep -q " passed" $O/t12.log || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof12 -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sample-steps 0 > $O/prof12.json 2> $O/prof12.err
f=$(find $O/prof12 -name "*kernel_stats.csv" | head -1); cp "$f" $O/r03b_train_kernel_stats.csv; rm -rf $O/prof12
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r03b_train_kernel_stats.csv')))
st=sum(int(r['Calls']) for r in rows if 'adamw_kernel' in r['Name'])
for r in rows:
    if any(k in r['Name'] for k in ('se_fwd','se_bwd','ca_z','ca_mix_kernel','ca_bwd')):
        print(f"{float(r['TotalDurationNs'])/st/1e6:7.3f} ms {int(r['Calls'])/st:5.1f}x {float(r['AverageNs'])/1e3:7.1f} us {r['Name'][:60]}")
PY
