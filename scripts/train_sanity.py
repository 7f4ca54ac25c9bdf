"""Overfit one synthetic batch for a few dozen steps (bf16 and fp32) and print the loss: a smoke test that the
whole step (halo kernels, split-K, bf16 shadow, batched re-packs, gradient arena) trains.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffusionmodel_amd as D

dev = "cuda:0"
for dtype in (torch.bfloat16, torch.float32):
    torch.manual_seed(0)
    net = D.ContextUnet(3, 64, 4, bottleneck_k=4, dtype=dtype)
    ddpm = D.DDPM(net, (1e-4, 0.02), 1000, dev, drop_prob=0.1)
    ddpm.train()
    ddpm.rng_seed = 5
    opt = D.FusedAdamW(ddpm.parameters(), lr=3e-4, weight_decay=1e-5, max_grad_norm=1.0)
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(16, 3, 64, 64, generator=g) * 2 - 1).to(dev)
    c = torch.randint(0, 4, (16,), generator=g).to(dev)
    am = torch.ones(16, 64, 64, device=dev)
    torch.manual_seed(11)
    ema, out = None, []
    for i in range(80):
        opt.zero_grad()
        loss = ddpm(x, c, am)
        loss.backward()
        opt.step()
        if i % 10 == 9:
            v = loss.item()
            ema = v if ema is None else 0.5 * ema + 0.5 * v
            out.append(f"{v:.4f}")
    print(str(dtype).split(".")[-1], "loss every 10 steps:", " ".join(out), "finite:", bool(torch.isfinite(opt.flat_p).all()))
