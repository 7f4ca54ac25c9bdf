#!/bin/bash
# Same-box A/B of the in-kernel pairwise fold of the halo weight gradient (DM_WGRAD_FOLD = "on[,max_splits[,partial_levels]]"): three
# alternations of scripts/bench_conv.py --what wgrad (dm_conv_wgrad incl. its reduce launch) per shape.
for only in "$@"; do
  for rep in 1 2 3; do
    for v in 0 1 "1,8,2" "1,8,3"; do
      echo -n "DM_WGRAD_FOLD=$v  "
      DM_WGRAD_FOLD=$v python scripts/bench_conv.py --what wgrad --only "$only" --iters 200 2>/dev/null | tail -1
    done
  done
done
