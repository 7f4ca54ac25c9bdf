#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 5 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -x -q -m gpu -k "embed or fc or linear or dense or context_unet or ddpm_forward" 2>&1 | tail -2
timeout -k 5 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --sample-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['families']['launches'])"
