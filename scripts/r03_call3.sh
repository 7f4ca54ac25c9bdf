#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_kernels.py -q -s -k "fused_coordattn or fused_se" > $O/t3.log 2>&1; echo "pytest rc $?" | tee -a $O/t3.log
tail -25 $O/t3.log
{
echo "== conv probe, one process"; DM_DEVICE_GUARD=0 timeout -k 5 120 python scripts/share_conv_probe.py solo 1500
echo "== conv probe, two processes"
( DM_DEVICE_GUARD=0 timeout -k 5 200 python scripts/share_conv_probe.py A 3000 > $O/sc_A.txt 2>&1 & ) ; DM_DEVICE_GUARD=0 timeout -k 5 200 python scripts/share_conv_probe.py B 3000 > $O/sc_B.txt 2>&1; sleep 6; cat $O/sc_A.txt $O/sc_B.txt
} > $O/probe3.txt 2>&1
cat $O/probe3.txt | grep -v amdgpu.ids
