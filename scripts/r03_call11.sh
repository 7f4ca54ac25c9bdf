#!/bin/bash
O=gpurun_out
run() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 0 "$@" 2> $O/b11.err | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']; f=r['families']; print('$*', d['ms_per_step'], 'kernel_ms', f['kernel_ms_per_step'])"; }
run
DM_DUMMY_SIDE=1 run
DM_DUMMY_SIDE=6 run
run --force-dp --buckets 6
run
