#!/bin/bash
# round-3 GPU call 2: co-residency LDS probe (two processes / two queues), fused SE chain test
set -o pipefail
O=gpurun_out
P=scripts/probes/lds_preempt.bin
{
for DMA in 0 1; do
echo "== dma=$DMA: process A 156 KiB x1/CU next to process B 8 KiB x4/CU"
( timeout -k 5 120 $P 156 20 150 A156 1 0 $DMA > $O/co_A.txt 2>&1 & ) ; timeout -k 5 120 $P 8 2 1500 B8 4 0 $DMA > $O/co_B.txt 2>&1 ; sleep 4 ; cat $O/co_A.txt $O/co_B.txt
echo "== dma=$DMA: process A 156 KiB next to process B 32 KiB x2/CU"
( timeout -k 5 120 $P 156 20 150 A156 1 0 $DMA > $O/co_A2.txt 2>&1 & ) ; timeout -k 5 120 $P 32 2 1500 B32 2 0 $DMA > $O/co_B2.txt 2>&1 ; sleep 4 ; cat $O/co_A2.txt $O/co_B2.txt
echo "== dma=$DMA: ONE process, 156 KiB on one stream + 8 KiB on a second stream"
timeout -k 5 120 $P 156 20 100 one156 1 8 $DMA
done
} > $O/probe2.txt 2>&1
cat $O/probe2.txt
python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_se" > $O/t2.log 2>&1; echo "pytest rc $?" | tee -a $O/t2.log
tail -15 $O/t2.log
