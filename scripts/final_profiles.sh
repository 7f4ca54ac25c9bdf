#!/bin/bash
# The round's measurement set on ONE box (usage on the GPU box: bash scripts/final_profiles.sh r04): bench line, same-box A/B toggles, rocprofv3 kernel stats (train / sample / cfg-5), PMC passes
O=gpurun_out
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 20 --warmup 5 > $O/${R}_bench_final.json 2> $O/${R}_bench_final.err; echo "bench rc $?"
ab() { env "$@" python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 0 --no-calibration --no-dp-probe 2> /dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']; print('$*', 'ms/step', d['ms_per_step'], 'halo frac', r['frac'], 'launches', r['families']['launches'], 'kernel ms', r['families']['kernel_ms_per_step'])"; }
{ ab A=default; ab DM_FUSED_CHAINS=0; ab DM_CONV_PERSIST=0; ab A=default; ab DM_FUSED_CHAINS=0 DM_CONV_PERSIST=0 DM_CONV_PACKTAP=0 DM_WGRAD_SKINNY=0 DM_SPLITK_INKERNEL=0; ab A=default; } > $O/${R}_ab_same_box.txt 2>&1
cat $O/${R}_ab_same_box.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pf1 -o t -- python3 bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --sample-steps 0 --no-calibration --no-dp-probe > /dev/null 2> $O/pf1.err
cp "$(find $O/pf1 -name '*kernel_stats.csv' | head -1)" $O/${R}_train_kernel_stats.csv; rm -rf $O/pf1
python3 scripts/kstats_families.py $O/${R}_train_kernel_stats.csv 0 $O/${R}_train_kernel_families.json > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pf2 -o t -- python3 scripts/sample_bench.py 40 > $O/pf2.log 2>&1
cp "$(find $O/pf2 -name '*kernel_stats.csv' | head -1)" $O/${R}_sample_kernel_stats.csv; rm -rf $O/pf2; grep "steps graph" $O/pf2.log
python3 bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline > $O/${R}_cfg5_bench.json 2> $O/${R}_cfg5_bench.err; echo "cfg5 rc $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pf3 -o t -- python3 bench.py --config cfg5 --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --sample-steps 0 > /dev/null 2> $O/pf3.err
cp "$(find $O/pf3 -name '*kernel_stats.csv' | head -1)" $O/${R}_cfg5_kernel_stats.csv; rm -rf $O/pf3
python3 scripts/kstats_families.py $O/${R}_cfg5_kernel_stats.csv 0 $O/${R}_cfg5_kernel_families.json | head -12
bash scripts/pmc_collect.sh ${R} > $O/pmc_${R}.log 2>&1
python3 scripts/pmc_traffic.py $O/pmc_${R}_fetch.csv $O/pmc_${R}_write.csv $O/${R}_pmc_hbm_traffic.json > /dev/null
python3 scripts/pmc_mfma.py $O/pmc_${R}_mfma.csv $O/${R}_pmc_mfma_busy.json > /dev/null
rm -f $O/pmc_${R}_fetch.csv $O/pmc_${R}_write.csv $O/pmc_${R}_mfma.csv
python3 scripts/step_timeline.py > /dev/null 2>&1
R_TAG=$R python3 - <<'PY'
import json
import os
R=os.environ.get('R_TAG','r04')
d=json.load(open(f'gpurun_out/{R}_bench_final.json')); r=d['roofline']
print('FINAL', d['value'], d['ms_per_step'], 'frac', r['frac'], 'step', r['step'], 'cpu', d['cpu_baseline']['value'], 'sample', d['sample'])
f=json.load(open(f'gpurun_out/{R}_train_kernel_families.json')); print({k:f[k] for k in ('steps','kernel_ms_per_step','launches_per_step','mfma_families_ms','non_mfma_ms')}, f['families'])
PY
