#!/bin/bash
# round-3 measurement set, one box: bench line, same-box A/B toggles, rocprofv3 kernel stats (train / sample / cfg-5), PMC passes
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 5 300 python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -k "packed_tap" 2>&1 | tail -1
python3 bench.py --steps 20 --warmup 5 > $O/r03_bench_final.json 2> $O/r03_bench_final.err; echo "bench rc $?"
ab() { env "$@" python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 0 2> /dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']; print('$*', 'ms/step', d['ms_per_step'], 'halo frac', r['frac'], 'launches', r['families']['launches'], 'kernel ms', r['families']['kernel_ms_per_step'])"; }
{ ab A=default; ab DM_FUSED_CHAINS=0; ab DM_CONV_PERSIST=0; ab DM_CONV_PACKTAP=0 DM_WGRAD_SKINNY=0; ab A=default; ab DM_SPLITK_INKERNEL=0; ab DM_FUSED_CHAINS=0 DM_CONV_PERSIST=0 DM_CONV_PACKTAP=0 DM_WGRAD_SKINNY=0 DM_SPLITK_INKERNEL=0; ab A=default; } > $O/r03_ab_same_box.txt 2>&1
cat $O/r03_ab_same_box.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pf1 -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sample-steps 0 > /dev/null 2> $O/pf1.err
cp "$(find $O/pf1 -name '*kernel_stats.csv' | head -1)" $O/r03_train_kernel_stats.csv; rm -rf $O/pf1
python3 scripts/kstats_families.py $O/r03_train_kernel_stats.csv 0 $O/r03_train_kernel_families.json > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pf2 -o t -- python3 scripts/sample_bench.py 40 > $O/pf2.log 2>&1
cp "$(find $O/pf2 -name '*kernel_stats.csv' | head -1)" $O/r03_sample_kernel_stats.csv; rm -rf $O/pf2; grep "steps graph" $O/pf2.log
python3 bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline > $O/r03_cfg5_bench.json 2> $O/r03_cfg5_bench.err; echo "cfg5 rc $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pf3 -o t -- python3 bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --sample-steps 0 > /dev/null 2> $O/pf3.err
cp "$(find $O/pf3 -name '*kernel_stats.csv' | head -1)" $O/r03_cfg5_kernel_stats.csv; rm -rf $O/pf3
python3 scripts/kstats_families.py $O/r03_cfg5_kernel_stats.csv 0 $O/r03_cfg5_kernel_families.json | head -12
bash scripts/pmc_collect.sh r03 > $O/pmc_r03.log 2>&1
python3 scripts/pmc_traffic.py $O/pmc_r03_fetch.csv $O/pmc_r03_write.csv $O/r03_pmc_hbm_traffic.json > /dev/null
python3 scripts/pmc_mfma.py $O/pmc_r03_mfma.csv $O/r03_pmc_mfma_busy.json > /dev/null
rm -f $O/pmc_r03_fetch.csv $O/pmc_r03_write.csv $O/pmc_r03_mfma.csv
python3 scripts/step_timeline.py > /dev/null 2>&1
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03_bench_final.json')); r=d['roofline']
print('FINAL', d['value'], d['ms_per_step'], 'frac', r['frac'], 'step', r['step'], 'cpu', d['cpu_baseline']['value'], 'sample', d['sample'])
f=json.load(open('gpurun_out/r03_train_kernel_families.json')); print({k:f[k] for k in ('steps','kernel_ms_per_step','launches_per_step','mfma_families_ms','non_mfma_ms')}, f['families'])
PY
