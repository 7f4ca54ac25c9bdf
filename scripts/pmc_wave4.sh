#!/bin/bash
# MFMA-pipe busy fraction and wave cycles of the 64x64 128->128 halo convolution per kernel form (rocprofv3 --pmc, kernel-trace only):
# persistent eight-wave, per-tile eight-wave, four-wave (compiler schedule / pinned prefetch / cross-tap timing experiment).
# usage (GPU box): bash scripts/pmc_wave4.sh  -> gpurun_out/pmc_wave4.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_wave4.txt
: > $out
for v in "1 0" "0 0" "0 1" "0 2"; do      # (+ "0 3" on a -DDM_HALO4_XTAP=1 build)
  set -- $v
  rm -rf gpurun_out/pmc_tmp
  DM_CONV_PERSIST=$1 DM_CONV_WAVE4=$2 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc_tmp -o p -- python3 scripts/bench_conv.py --what fwd --only "64^2 128->128 3x3" --iters 200 > gpurun_out/pmc_wave4_run.log 2>&1 || { echo "failed $v" >> $out; tail -5 gpurun_out/pmc_wave4_run.log >> $out; continue; }
  f=$(find gpurun_out/pmc_tmp -name "*counter_collection.csv" | head -1)
  python3 - "$f" "DM_CONV_PERSIST=$1 DM_CONV_WAVE4=$2" >> $out <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "halo" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
    d = by[r["Dispatch_Id"]]
    d[r["Counter_Name"]] = float(r["Counter_Value"])
    d["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    d["name"] = r["Kernel_Name"][:44]
ds = list(by.values())[len(by) // 3:]
n = len(ds)
avg = lambda k: sum(d.get(k, 0.0) for d in ds) / n
print(f"{sys.argv[2]:34s} {ds[0]['name']:44s} {avg('ns') / 1e3:7.2f} us  MFMA busy {avg('SQ_VALU_MFMA_BUSY_CYCLES') / max(4 * avg('SQ_BUSY_CU_CYCLES'), 1):.3f}  "
      f"wave cycles {avg('SQ_WAVE_CYCLES'):.3e}  WAIT_INST_ANY {avg('SQ_WAIT_INST_ANY') / max(avg('SQ_WAVE_CYCLES'), 1):.3f}  WAIT_ANY {avg('SQ_WAIT_ANY') / max(avg('SQ_WAVE_CYCLES'), 1):.3f}")
PY
done
rm -rf gpurun_out/pmc_tmp
cat $out
