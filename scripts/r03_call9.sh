#!/bin/bash
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $O/prof_r03
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03 -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sample-steps 0 > $O/prof_r03_bench.json 2> $O/prof_r03_bench.err
echo "rocprof bench rc $?"
f=$(find $O/prof_r03 -name "*kernel_stats.csv" | head -1); cp "$f" $O/r03_train_kernel_stats.csv; rm -rf $O/prof_r03
python3 scripts/kstats_families.py $O/r03_train_kernel_stats.csv 27 $O/r03_train_kernel_families.json | head -40
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03s -o t -- python3 scripts/sample_bench.py > $O/prof_r03_sample.log 2>&1
echo "rocprof sample rc $?"; tail -2 $O/prof_r03_sample.log
f=$(find $O/prof_r03s -name "*kernel_stats.csv" | head -1); cp "$f" $O/r03_sample_kernel_stats.csv; rm -rf $O/prof_r03s
