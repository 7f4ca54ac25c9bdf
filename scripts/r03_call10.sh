#!/bin/bash
O=gpurun_out
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 0 "$@" 2> $O/b10.err | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']; print('$*', d['ms_per_step'], 'launches', r['families']['launches'], 'optimiser ms', r['families']['ms'].get('optimiser'), 'host', d['config']['host_ms_one_step_idle_queue'])"; }
run
run --force-dp --buckets 6
run --force-dp --buckets 3
run --force-dp --buckets 1
run
DM_DP_NOCOMM=1 run --force-dp --buckets 6
