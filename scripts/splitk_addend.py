"""In-kernel vs two-launch split-K through the forked-gradient addend path (dgrad epilogue adds a stashed gradient), and with BN statistics."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionmodel_amd import _lib as L, ops as o
lib = L.load(); dev = "cuda:0"
g = torch.Generator().manual_seed(7)
for (B, H, C) in ((2, 64, 256), (2, 32, 256), (2, 16, 512), (64, 16, 256), (4, 8, 1024)):
    N = C
    w = torch.nn.Parameter((torch.randn(N, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last))
    x = torch.randn(B, H, H, C, generator=g).to(dev).bfloat16()
    gy = torch.randn(B, H, H, N, generator=g).to(dev).bfloat16()
    class Hd:
        weight, bias = w, None
    bn = torch.nn.BatchNorm2d(N).to(dev).train()
    sd = {k: v.clone() for k, v in bn.state_dict().items()}
    res = {}
    for ink in (1, 0):
        lib.dm_set_splitk_inkernel(ink)
        for with_bn in (False, True):
            bn.load_state_dict(sd)
            xg = x.clone().requires_grad_(True)
            fork = o.GradFork()
            spec = o.ConvSpec(3, 3, 1, 1, o.ACT_GELU if with_bn else o.ACT_NONE, bn if with_bn else None)
            y = o.conv_bn_act(xg, None, Hd, bn if with_bn else None, spec, fork)
            loss = (y.float() * gy.float()).sum() + ((fork.second(xg).float() + 0.0 * y.float()) * 0.37).sum()
            loss.backward()
            res[(ink, with_bn)] = (y.detach().float(), xg.grad.float())
    for with_bn in (False, True):
        a, b = res[(1, with_bn)], res[(0, with_bn)]
        dy, dx = (a[0] - b[0]).abs(), (a[1] - b[1]).abs()
        print(f"B{B} {H}x{H} C{C} bn={with_bn}: y differs in {int((dy > 0).sum())} (max {float(dy.max()):.3g}); dx differs in {int((dx > 0).sum())} of {dx.numel()} (max {float(dx.max()):.3g}, |dx| max {float(b[1].abs().max()):.3g})", flush=True)
lib.dm_set_splitk_inkernel(1)
