#!/bin/bash
# PMC passes over the bench command (separate rocprofv3 runs per counter group, --kernel-trace only: gpurun refuses pmc + other traces).
# usage (on the GPU box): bash scripts/pmc_collect.sh <tag>      -> gpurun_out/pmc_<tag>_{fetch,write,mfma}.csv (counter_collection tables)
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python3 bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --sample-steps 0 --no-calibration --no-dp-probe"
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctr=${pass#*:}
  rm -rf gpurun_out/pmc_tmp
  DM_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc_tmp -o p -- $CMD > gpurun_out/pmc_${TAG}_${name}.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/pmc_${TAG}_${name}.log; continue; }
  f=$(find gpurun_out/pmc_tmp -name "*counter_collection.csv" | head -1)
  python3 - "$f" gpurun_out/pmc_${TAG}_${name}.csv <<'PY'
import csv, sys
# keep only what the summaries need: kernel name, counter, value, dispatch id (the raw table is tens of MB)
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ("Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp")
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=[k for k in keep if k in rows[0]])
    w.writeheader()
    for r in rows:
        r["Kernel_Name"] = r["Kernel_Name"][:80]
        w.writerow({k: r[k] for k in w.fieldnames})
print(sys.argv[2], len(rows), "rows")
PY
  echo "pass $name done"
done
rm -rf gpurun_out/pmc_tmp
