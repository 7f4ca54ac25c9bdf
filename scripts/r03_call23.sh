#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 5 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "packed_tap" 2>&1 | grep -v "^E  \|^$" | tail -8 || exit 1
for v in 1 0; do echo "packtap=$v"; DM_CONV_PACKTAP=$v timeout -k 5 200 python scripts/bench_conv.py --what fwd --iters 20 --only "head" 2>&1 | grep head; done
for rep in 1 2 3; do for v in 1 0; do
  DM_CONV_PACKTAP=$v timeout -k 5 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('packtap=$v', d['ms_per_step'], d['roofline']['families']['ms'].get('conv'), d['sample']['steps_per_s'])" || exit 1
done; done
