#!/bin/bash
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "splitk or halo_kernel_exact or four_tap" --timeout 120 > $O/t14.log 2>&1; echo "pytest rc $?"; tail -3 $O/t14.log
grep -q " passed" $O/t14.log || exit 1
for V in 1 0 1 0; do
DM_SPLITK_INKERNEL=$V python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 0 2> $O/b14.err | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']; f=r['families']; print('inkernel=$V', d['ms_per_step'], 'launches', f['launches'], 'conv ms', f['ms']['conv'], 'loss', d['loss'])"
done
