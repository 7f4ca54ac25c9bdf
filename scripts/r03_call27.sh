#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "dp_product or overlapped or driver_engine or two_rank" 2>&1 | tail -2
for rep in 1 2 3; do
  for mode in "plain" "dp side=1" "dp side=0"; do
    case "$mode" in
      plain) F=""; E="DM_DP_SIDE_STREAM=1";;
      "dp side=1") F="--force-dp"; E="DM_DP_SIDE_STREAM=1";;
      "dp side=0") F="--force-dp"; E="DM_DP_SIDE_STREAM=0";;
    esac
    env $E MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout -k 5 300 python bench.py $F --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'])"
  done
done
