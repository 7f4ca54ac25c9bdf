#!/bin/bash
# Same-box A/B of two library builds on single weight-gradient shapes (dm_conv_wgrad + its reduce launch): libdm_amd_prev.so vs libdm_amd.so.
# usage (GPU box): bash scripts/ab_wgrad.sh "<shape substring>" [more substrings]
for only in "$@"; do
  for rep in 1 2 3; do
    for v in libdm_amd_prev.so libdm_amd.so; do
      echo -n "$v  "
      DM_LIB_PATH=$GRAFT_REPO_ROOT/diffusionmodel_amd/$v python scripts/bench_conv.py --what wgrad --only "$only" --iters 200 2>/dev/null | tail -1
    done
  done
done
