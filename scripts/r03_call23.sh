#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 5 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "packed_tap" 2>&1 | tail -2
timeout -k 5 300 python scripts/_dbg_sk.py
for rep in 1 2 3; do for v in 1 0; do
  DM_WGRAD_SKINNY=$v timeout -k 5 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('skinny=$v', d['ms_per_step'], d['roofline']['families']['ms'].get('wgrad'))" || exit 1
done; done
