#!/bin/bash
# Same-box A/B of an environment toggle on the whole train step: three alternations of bench.py (median of 5 regions each).
# usage (GPU box): bash scripts/ab_env.sh "DM_LAZY_ZERO=0" "DM_LAZY_ZERO=1" [-- extra bench args]
A="$1"; B="$2"; shift 2; [ "$1" = "--" ] && shift
for rep in 1 2 3; do
  for v in "$A" "$B"; do
    env $v python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 0 --no-calibration --no-dp-probe "$@" 2> /dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']; f=r['families']
print('$v', 'ms/step', d['ms_per_step'], 'min', d['repeats']['ms_per_step_min'], 'launches', f['launches'], 'kernel ms', f['kernel_ms_per_step'], 'optimiser', f['ms'].get('optimiser'), 'wgrad', f['ms'].get('wgrad'), 'torch', f['ms'].get('torch'))"
  done
done
