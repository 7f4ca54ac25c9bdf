#!/bin/bash
# Same-box A/B of two library builds: diffusionmodel_amd/libdm_amd_prev.so (copy of an earlier build) vs libdm_amd.so, three alternating runs.
# usage (on the GPU box): bash scripts/ab_lib.sh
for rep in 1 2 3; do
for v in libdm_amd_prev.so libdm_amd.so; do
  DM_LIB_PATH=$GRAFT_REPO_ROOT/diffusionmodel_amd/$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
done; done
