"""Per-family summary of a rocprofv3 *_kernel_stats.csv of `bench.py --sample-steps 0 --no-cpu-baseline` (families of bench.py):
ms per step and launches per step; MFMA families (conv, wgrad) vs everything else.   usage: kstats_families.py <csv> <steps> [out.json]"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import family_of

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
if steps <= 0:                                 # executed steps = launches of the optimiser kernel
    steps = float(sum(int(r["Calls"]) for r in rows if "adamw_kernel" in r["Name"]))
fam, sub = {}, {}
for r in rows:
    f = family_of(r["Name"])
    ms, n = float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps
    a = fam.setdefault(f, [0.0, 0.0])
    a[0] += ms; a[1] += n
    key = r["Name"].split("(")[0][:60]
    for k in ("conv3x3_halo_pkernel", "conv3x3_halo_kernel", "conv3x3_packtap", "conv3x3_narrow", "conv_tap4_halo", "conv_pw_kernel", "conv_igemm2", "splitk_epilogue", "wgrad3x3_halo", "wgrad3x3_skinny", "wgrad_skinny_reduce", "wgrad_reduce",
              "wgrad_pw", "conv_wgrad2", "bn_bwd_apply", "bn_bwd_reduce", "bn_act_fwd", "adamw", "sumsq", "pack_multi", "se_fwd", "se_bwd", "ca_z_", "ca_mix",
              "ca_bwd_mix", "ca_bwd_z", "strip_reduce", "scale_res", "ca_pix", "ca_gate", "dense_", "upcat", "gn_", "film"):
        if k in r["Name"]:
            key = k
            break
    b = sub.setdefault((f, key), [0.0, 0.0])
    b[0] += ms; b[1] += n
tot = sum(v[0] for v in fam.values())
mfma = sum(v[0] for k, v in fam.items() if k in ("conv", "wgrad"))
out = {"source": os.path.basename(sys.argv[1]), "steps": steps, "kernel_ms_per_step": round(tot, 3), "launches_per_step": round(sum(v[1] for v in fam.values()), 1),
       "mfma_families_ms": round(mfma, 3), "non_mfma_ms": round(tot - mfma, 3),
       "families": {k: {"ms": round(v[0], 3), "launches": round(v[1], 1)} for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])},
       "kernels": {f"{f}/{k}": {"ms": round(v[0], 3), "launches": round(v[1], 1)} for (f, k), v in sorted(sub.items(), key=lambda kv: -kv[1][0]) if v[0] >= 0.02}}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
