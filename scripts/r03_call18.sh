#!/bin/bash
# same-box A/B of weight-gradient builds (DM_WGRAD_DMA_POS / DM_WGRAD_ASM_READS): kernel tests per build, then us per launch
cd $GRAFT_REPO_ROOT
for v in diffusionmodel_amd/libdm_amd_wpos*.so; do
  DM_LIB_PATH=$GRAFT_REPO_ROOT/$v timeout -k 5 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wgrad or weight_grad or exact" 2>&1 | tail -1
done
for rep in 1 2; do
for v in diffusionmodel_amd/libdm_amd_wpos*.so; do
  echo "== $(basename $v) rep $rep"
  DM_LIB_PATH=$GRAFT_REPO_ROOT/$v timeout -k 5 200 python scripts/bench_conv.py --what wgrad --iters 20 2>&1 | grep "3x3" | grep -v "head\|stem\|cfg-5" || exit 1
done; done
