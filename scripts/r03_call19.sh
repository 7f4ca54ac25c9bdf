#!/bin/bash
# wgrad halo kernel with the next tile's LDS-DMA issued in k-step 1 (between MFMAs): kernel tests with the variant build, then whole-step A/B
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/diffusionmodel_amd
V=libdm_amd_wpos0x0f00_1.so
DM_LIB_PATH=$L/$V timeout -k 5 500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wgrad or weight_grad or exact or conv" > gpurun_out/t_wpos.log 2>&1 || { tail -20 gpurun_out/t_wpos.log; exit 1; }
tail -2 gpurun_out/t_wpos.log
for rep in 1 2 3; do
for v in libdm_amd.so $V; do
  DM_LIB_PATH=$L/$v timeout -k 5 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['frac'], d['roofline']['families']['ms'].get('wgrad'))" || exit 1
done; done
