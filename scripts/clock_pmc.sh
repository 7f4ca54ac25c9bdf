#!/bin/bash
# Clock held under the halo convolution on random vs all-zero operands: GRBM_GUI_ACTIVE cycles per launch / launch duration (rocprofv3 --pmc,
# kernel-trace only).  usage (GPU box): bash scripts/clock_pmc.sh  -> gpurun_out/clock_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/clock_pmc.txt
: > $out
for mode in random zeros; do
  for shape in "64^2 128->128 3x3" "16^2 512->512 3x3"; do
    rm -rf gpurun_out/pmc_tmp
    extra=""; [ $mode = zeros ] && extra="--zeros"
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_tmp -o p -- python3 scripts/bench_conv.py --what fwd --only "$shape" --iters 300 $extra > gpurun_out/clock_pmc_run.log 2>&1 || { echo "failed $mode $shape" >> $out; tail -5 gpurun_out/clock_pmc_run.log >> $out; continue; }
    f=$(find gpurun_out/pmc_tmp -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$mode" "$shape" >> $out <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "halo" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
    d = by[r["Dispatch_Id"]]
    d[r["Counter_Name"]] = float(r["Counter_Value"])
    d["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
ds = list(by.values())[len(by) // 3:]          # the steady part of the run
n = len(ds)
avg = lambda k: sum(d.get(k, 0.0) for d in ds) / n
us = avg("ns") / 1e3
print(f"{sys.argv[2]:6s} {sys.argv[3]:22s} launches {n:4d}  {us:7.2f} us  GRBM_GUI_ACTIVE {avg('GRBM_GUI_ACTIVE'):10.0f} cyc -> {avg('GRBM_GUI_ACTIVE') / (us * 1e3):.3f} GHz  "
      f"MFMA busy / CU busy {avg('SQ_VALU_MFMA_BUSY_CYCLES') / max(avg('SQ_BUSY_CU_CYCLES'), 1):.3f}  wave cycles {avg('SQ_WAVE_CYCLES'):.3e}")
PY
  done
done
rm -rf gpurun_out/pmc_tmp
cat $out
