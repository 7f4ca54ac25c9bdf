#!/bin/bash
# round-3 GPU call 1: LDS-preemption probe, KFD sysfs visibility, full GPU test suite, bench line
set -o pipefail
O=gpurun_out
P=scripts/probes/lds_preempt.bin
{
echo "== solo 156 KiB"; timeout -k 5 60 $P 156 20 40 solo
echo "== two processes, 156 KiB each"
( timeout -k 5 120 $P 156 20 150 A > $O/lds_A.txt 2>&1 & ) ; timeout -k 5 120 $P 156 20 150 B > $O/lds_B.txt 2>&1 ; sleep 4 ; cat $O/lds_A.txt $O/lds_B.txt
echo "== two processes, 64 KiB each"
( timeout -k 5 120 $P 64 20 150 A64 > $O/lds_A64.txt 2>&1 & ) ; timeout -k 5 120 $P 64 20 150 B64 > $O/lds_B64.txt 2>&1 ; sleep 4 ; cat $O/lds_A64.txt $O/lds_B64.txt
echo "== two processes, 96 KiB each"
( timeout -k 5 120 $P 96 20 150 A96 > $O/lds_A96.txt 2>&1 & ) ; timeout -k 5 120 $P 96 20 150 B96 > $O/lds_B96.txt 2>&1 ; sleep 4 ; cat $O/lds_A96.txt $O/lds_B96.txt
echo "== kfd sysfs"; ls -la /sys/class/kfd/kfd/proc 2>&1 | head; for d in /sys/class/kfd/kfd/proc/*; do echo $d; ls $d 2>&1 | head -20; cat $d/pasid 2>&1; ls $d/queues 2>&1 | head; done 2>&1 | head -60
id; ls -la /tmp | head -5
} > $O/probe1.txt 2>&1
echo probe done
python -m pytest tests -m gpu -x -q -s > $O/t1.log 2>&1; echo "pytest rc $?" | tee -a $O/t1.log
tail -5 $O/t1.log
python bench.py --steps 20 --warmup 5 > $O/bench_r03_v1.json 2> $O/bench_r03_v1.err; echo "bench rc $?"
tail -c 600 $O/bench_r03_v1.json
