import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_lib as PL
u = PL.unet_case(torch.float16)
for m in u:
    r = u[m]
    print(m, "mse", r["eps_mse_vs_ref64"], "ref", r["ref_autocast_bf16_mse"], "max", r["eps_maxabs_vs_ref64"], r["ref_autocast_bf16_maxabs"], "loss", r["loss"], r["loss_ref64"])
    print("  worst norm", max(abs(v["norm_rel_err"]) for v in r["grads"].values()), "ref", max(abs(v["norm_rel_err"]) for v in r["ref_autocast_bf16_grads"].values()))
    print("  worst 1-cos ratio", max(r["grads"][c]["one_minus_cos"] / max(r["ref_autocast_bf16_grads"][c]["one_minus_cos"], 1e-12) for c in r["grads"]),
          max(v["one_minus_cos"] for v in r["grads"].values()))
d = PL.ddpm_case(torch.float16)
print({k: v for k, v in d["train"].items() if "grads" not in k}, d["eval"])
print("  ddpm worst 1-cos ratio", max(d["train"]["grads"][c]["one_minus_cos"] / max(d["train"]["ref_autocast_bf16_grads"][c]["one_minus_cos"], 1e-12) for c in d["train"]["grads"]))
t = PL.train3_case(torch.float16)
print({k: t[k] for k in ("losses", "losses_ref", "grad_norms", "grad_norms_ref", "loss_rel_err", "grad_norm_rel_err")})
