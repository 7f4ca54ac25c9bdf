#!/usr/bin/env python3
"""Write the parity measurements of tests/parity_lib.py (HIP bf16 / fp32 against the reference's float64 and bf16-autocast runs,
the 3-optimiser-step trajectory, the F=128 case, the conv+BN kernel) as JSON.  GPU only.

    python scripts/parity_report.py gpurun_out/r02_parity.json      # then copy to profiles/r02_parity.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_lib  # noqa: E402

out = parity_lib.measure_all()
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r02_parity.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
with open(path, "w") as f:
    json.dump(out, f, indent=1)


def show(d, ind=0):
    for k, v in d.items():
        if isinstance(v, dict) and any(isinstance(x, dict) for x in v.values()):
            print(" " * ind + str(k))
            show(v, ind + 2)
        else:
            print(" " * ind + f"{k}: {v}")


show(out)
