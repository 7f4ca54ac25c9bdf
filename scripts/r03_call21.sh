#!/bin/bash
# spread of the fp16 three-step test: current build vs the build of two commits ago, three processes each
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in diffusionmodel_amd/libdm_amd.so prev_build/diffusionmodel_amd/libdm_amd.so; do
  echo "== $v rep $rep"
  DM_LIB_PATH=$GRAFT_REPO_ROOT/$v timeout -k 5 300 python -m pytest tests/test_gpu_bf16.py -q -m gpu -s -k "three_optimiser_steps" 2>&1 | grep "fp16 train3\|passed\|failed"
done; done
