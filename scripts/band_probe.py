"""Print the per-child gradient-norm error distributions of tests/parity_lib.py:f128_b2_band_case (GPU)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_lib as PL
r = PL.f128_b2_band_case()
print(json.dumps(r, indent=1))
