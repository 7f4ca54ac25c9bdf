#!/bin/bash
O=gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q --timeout 300 > $O/t13.log 2>&1; echo "pytest rc $?"; tail -3 $O/t13.log
grep -q " passed" $O/t13.log || exit 1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 0 2> $O/b13.err | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']; f=r['families']; print(d['ms_per_step'], 'kernel_ms', f['kernel_ms_per_step'], f['ms'])"
