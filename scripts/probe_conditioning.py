"""Diagnostic (GPU): how sensitive is a scalar gradient (ca2.gamma_h, train mode) to 1e-7-level input noise?"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import synth
import diffusionmodel_amd as D
G = os.path.join(ROOT, "tests", "golden")
SCHEMA = json.load(open(os.path.join(G, "schema.json")))
tag = "unet32_64"
g = np.load(os.path.join(G, tag + ".npz"))
x = synth.synth_input(tag + ".x", (2, 3, 64, 64))
c, t, mk = torch.tensor(g["c"]), torch.tensor(g["t"]), torch.tensor(g["ctx_mask"])
names = ["ca2.gamma_h", "ca1.alpha", "ca2.gamma_w", "ca3.gamma_h", "out.3.weight", "down1.ch_adjust.weight"]
res = []
for trial in range(4):
    net = D.ContextUnet(3, 32, 4, bottleneck_k=4)
    net.load_state_dict({k: synth.synth_tensor(k, tuple(s)) for k, s in SCHEMA[tag]})
    net = net.to("cuda:0").train()
    xx = x * (1 + 1e-7 * trial * torch.randn_like(x))
    eps = net(xx.cuda(), c.cuda(), t.cuda(), mk.cuda())
    (eps * synth.synth_input(tag + ".probe", tuple(eps.shape)).cuda()).mean().backward()
    P = dict(net.named_parameters())
    res.append({n: P[n].grad.double().cpu().numpy().copy() for n in names})
for n in names:
    base = res[0][n]
    ref = g["train.g." + n] if ("train.g." + n) in g.files else None
    spread = max(np.abs(r[n] - base).max() for r in res[1:]) / max(np.abs(base).max(), 1e-30)
    msg = f"{n}: rel spread under 1e-7 input noise = {spread:.2e}"
    if ref is not None:
        msg += f"; rel err vs reference fp32 = {np.abs(base - ref).max() / np.abs(ref).max():.2e}"
    print(msg)
