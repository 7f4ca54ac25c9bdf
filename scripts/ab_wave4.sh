#!/bin/bash
# Same-box A/B of the four-wave form of the 3x3 halo kernel (DM_CONV_WAVE4 = 0 / 1 / 2) against the eight-wave per-tile kernel
# (DM_CONV_PERSIST=0) and the persistent kernel (default), three alternations per shape.   usage: bash scripts/ab_wave4.sh "<shape>" ...
for only in "$@"; do
  for rep in 1 2 3; do
    for v in "DM_CONV_PERSIST=1 DM_CONV_WAVE4=0" "DM_CONV_PERSIST=0 DM_CONV_WAVE4=0" "DM_CONV_PERSIST=0 DM_CONV_WAVE4=1" "DM_CONV_PERSIST=0 DM_CONV_WAVE4=2"; do      # (+ "DM_CONV_PERSIST=0 DM_CONV_WAVE4=3" on a -DDM_HALO4_XTAP=1 build)
      echo -n "$v  "
      env $v python scripts/bench_conv.py --what fwd --only "$only" --iters 200 2>/dev/null | tail -1
    done
  done
done
