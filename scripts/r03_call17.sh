#!/bin/bash
# LDS-DMA placement in the halo kernels (persistent form included): conv tests with the variant build, then whole-step same-box A/B
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/diffusionmodel_amd
DM_LIB_PATH=$L/libdm_amd_pos567.so timeout -k 5 500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or halo or persist or splitk" > gpurun_out/t_pos567.log 2>&1 || { tail -20 gpurun_out/t_pos567.log; exit 1; }
tail -2 gpurun_out/t_pos567.log
for rep in 1 2 3; do
for v in libdm_amd.so libdm_amd_pos567.so libdm_amd_pos357.so; do
  DM_LIB_PATH=$L/$v timeout -k 5 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['frac'])" || exit 1
done; done
