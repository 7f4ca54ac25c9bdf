#!/bin/bash
cd $GRAFT_REPO_ROOT
DM_DEEP_BN128=1 timeout -k 5 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or splitk or tap4 or 4x4" 2>&1 | tail -2
for rep in 1 2 3; do for v in 1 0; do
  DM_DEEP_BN128=$v timeout -k 5 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('deep=$v', d['ms_per_step'], d['roofline']['families']['ms'].get('conv'), d['sample']['steps_per_s'])" || exit 1
done; done
