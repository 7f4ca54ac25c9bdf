"""In-kernel vs two-launch split-K on REAL-valued data: how far apart are the two (fp32 summation order only?)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionmodel_amd import _lib as L, ops as o
lib = L.load(); dev = "cuda:0"
g = torch.Generator().manual_seed(7)
sp = o.ConvSpec(3, 3, 1, 1)
for (B, H, C, N) in ((64, 16, 256, 256), (64, 16, 1024, 256), (64, 8, 1024, 1024), (64, 8, 2048, 512)):
    w = torch.nn.Parameter((torch.randn(N, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last))
    x = torch.randn(B, H, H, C, generator=g).to(dev).bfloat16()
    gy = torch.randn(B, H, H, N, generator=g).to(dev).bfloat16()
    class Hd:
        weight, bias = w, None
    # fp32 reference of the same bf16 operands
    xr = x.float().permute(0, 3, 1, 2); wr = w.detach().bfloat16().float()
    yr = torch.nn.functional.conv2d(xr, wr, padding=1).permute(0, 2, 3, 1)
    out = {}
    for ink in (1, 0, 1):
        lib.dm_set_splitk_inkernel(ink)
        xg = x.clone().requires_grad_(True)
        y = o.conv_bn_act(xg, None, Hd, None, sp)
        y.backward(gy)
        out.setdefault(ink, []).append((y.detach().float(), xg.grad.float()))
    a, b, a2 = out[1][0], out[0][0], out[1][1]
    def cmp(u, v): d = (u - v).abs(); return int((d > 0).sum()), float(d.max()), float(d.mean())
    err = lambda u: float(((u - yr) ** 2).mean().sqrt())
    print(f"B{B} {H}x{H} C{C} N{N}: y in-kernel vs two-launch {cmp(a[0], b[0])} of {a[0].numel()}; dx {cmp(a[1], b[1])}; in-kernel run 1 vs run 2: y {cmp(a[0], a2[0])} dx {cmp(a[1], a2[1])};"
          f" rms err vs fp32 conv: in-kernel {err(a[0]):.3e} two-launch {err(b[0]):.3e}", flush=True)
lib.dm_set_splitk_inkernel(1)
