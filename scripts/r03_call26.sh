#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do echo rmw=$v; DM_WGRAD_RMW=$v timeout -k 5 200 python scripts/bench_conv.py --what wgrad --iters 20 --only "8^2 1024->1024 4x4s2" 2>&1 | grep 4x4; done
for rep in 1 2 3; do for v in 1 0; do
DM_WGRAD_RMW=$v timeout -k 5 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rmw=$v', d['ms_per_step'], d['roofline']['families']['ms']['wgrad'])"
done; done
