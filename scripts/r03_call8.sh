#!/bin/bash
# rocprofv3 kernel stats of the bench command + PMC passes (separate runs)
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $O/prof_r03
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03 -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_r03_bench.json 2> $O/prof_r03_bench.err
echo "rocprof bench rc $?"
f=$(find $O/prof_r03 -name "*kernel_stats.csv" | head -1); echo "stats: $f"
cp "$f" $O/r03_train_kernel_stats.csv
python3 scripts/kstats.py $O/r03_train_kernel_stats.csv 26 25
rm -rf $O/prof_r03
bash scripts/pmc_collect.sh r03
python3 scripts/pmc_traffic.py $O/pmc_r03_fetch.csv $O/pmc_r03_write.csv $O/r03_pmc_hbm_traffic.json && head -c 1500 $O/r03_pmc_hbm_traffic.json
python3 scripts/pmc_mfma.py $O/pmc_r03_mfma.csv $O/r03_pmc_mfma_busy.json && head -c 1800 $O/r03_pmc_mfma_busy.json
rm -f $O/pmc_r03_fetch.csv $O/pmc_r03_write.csv $O/pmc_r03_mfma.csv
