import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import parity_lib as PL
from diffusionmodel_amd import _lib as L
lib = L.load()
for ink in (1, 0, 1, 0):
    lib.dm_set_splitk_inkernel(ink)
    r = PL.f128_b2_case(torch.bfloat16, modes=("eval",))["eval"]
    g = r["grad_norm_rel_err"]
    print("inkernel", ink, "eps mse", f"{r['eps_mse_vs_ref64']:.3e}", {k: round(v, 4) for k, v in g.items() if abs(v) > 0.03}, "ref worst", round(max(abs(v) for v in r["ref_autocast_grad_norm_rel_err"].values()), 4),
          {k: round(v, 4) for k, v in r["ref_autocast_grad_norm_rel_err"].items() if k.startswith("ca")}, flush=True)
