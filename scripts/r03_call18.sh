#!/bin/bash
# same-box A/B of the LDS-DMA placement inside the halo weight-gradient kernel (DM_WGRAD_DMA_POS builds)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in diffusionmodel_amd/libdm_amd.so diffusionmodel_amd/libdm_amd_wpos*.so; do
  echo "== $(basename $v) rep $rep"
  DM_LIB_PATH=$GRAFT_REPO_ROOT/$v timeout -k 5 200 python scripts/bench_conv.py --what wgrad --iters 20 2>&1 | grep "3x3" | grep -v "head\|stem\|cfg-5" || exit 1
done; done
