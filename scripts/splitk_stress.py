"""Stress of the in-kernel split-K hand-off (sc1 stores / counter / sc1 loads, igemm_dev.h splitk_last_arriver) for STALE reads: fresh
random integer data in every repetition, several layer shapes in rotation (they share the workspace), a weight-gradient launch in
between (plain stores / loads on the same workspace); the in-kernel result must equal the two-launch result of the same repetition
bit for bit — forward and input gradient.   python scripts/splitk_stress.py [reps]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionmodel_amd import _lib as L, ops as o

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
lib = L.load()
dev = "cuda:0"
g = torch.Generator().manual_seed(5)
shapes = ((64, 8, 1024, 1024), (64, 8, 2048, 512), (64, 16, 256, 256), (64, 16, 1024, 256), (64, 8, 512, 512))
layers = []
for (B, H, C, N) in shapes:
    w = torch.nn.Parameter((torch.randint(-1, 2, (N, C, 3, 3), generator=g).float() * (torch.rand(N, C, 3, 3, generator=g) < 0.05).float())
                           .to(dev).contiguous(memory_format=torch.channels_last))
    layers.append((B, H, C, N, w))
sp = o.ConvSpec(3, 3, 1, 1)
bad = {s: 0 for s in shapes}
first = None
for r in range(reps):
    for (B, H, C, N, w) in layers:
        x = torch.randint(-2, 3, (B, H, H, C), generator=g).float().to(dev).bfloat16()
        gy = torch.randint(-1, 2, (B, H, H, N), generator=g).float().to(dev).bfloat16()

        class Hd:
            weight, bias = w, None
        res = []
        for ink in (1, 0):
            lib.dm_set_splitk_inkernel(ink)
            xg = x.clone().requires_grad_(True)
            w.grad = None
            y = o.conv_bn_act(xg, None, Hd, None, sp)
            y.backward(gy)                              # input gradient (split-K too) + weight gradient (plain workspace traffic)
            res.append((y.detach().clone(), xg.grad.clone()))
        ok = torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        if not ok:
            bad[(B, H, C, N)] += 1
            if first is None:
                d = (res[0][0].float() - res[1][0].float()).abs()
                d2 = (res[0][1].float() - res[1][1].float()).abs()
                first = (r, (B, H, C, N), int((d > 0).sum()), float(d.max()), int((d2 > 0).sum()), float(d2.max()))
lib.dm_set_splitk_inkernel(1)
for s_, n in bad.items():
    print(f"B{s_[0]} {s_[1]}x{s_[1]} C{s_[2]} N{s_[3]}: {n} of {reps} repetitions differ from the two-launch result", flush=True)
print("first mismatch (rep, shape, #y elements, max |dy|, #dx elements, max |ddx|):", first)
print("splitk stress", "OK" if not any(bad.values()) else "FAILED", flush=True)
