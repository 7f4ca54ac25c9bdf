#!/bin/bash
# Same-box A/B of two library builds on single conv shapes: libdm_amd_prev.so vs libdm_amd.so, three alternations of scripts/bench_conv.py.
# usage (GPU box): bash scripts/ab_conv.sh "<shape substring>" [more substrings]
for only in "$@"; do
  for rep in 1 2 3; do
    for v in libdm_amd_prev.so libdm_amd.so; do
      echo -n "$v  "
      DM_LIB_PATH=$GRAFT_REPO_ROOT/diffusionmodel_amd/$v python scripts/bench_conv.py --what fwd --only "$only" --iters 200 | tail -1
    done
  done
done
