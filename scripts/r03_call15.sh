#!/bin/bash
O=gpurun_out
timeout -k 5 300 python scripts/f128_spread.py 2>&1 | grep -v amdgpu | cut -c1-200
timeout -k 10 600 python -m pytest tests -m gpu -x -q --timeout 300 > $O/t15.log 2>&1; echo "pytest rc $?"; tail -3 $O/t15.log
grep -q " passed" $O/t15.log || exit 1
for V in 1 0 1 0; do
DM_SPLITK_INKERNEL=$V python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 0 2> $O/b15.err | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']; f=r['families']; print('inkernel=$V', d['ms_per_step'], 'launches', f['launches'], 'conv ms', f['ms']['conv'], 'loss', d['loss'])"
done
