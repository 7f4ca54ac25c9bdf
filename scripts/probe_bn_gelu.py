"""Accuracy of the 16-bit BN + GELU streaming kernels against a float64 evaluation of the same formulas."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionmodel_amd import _lib as L
from diffusionmodel_amd.ops import call, ptr

dev = "cuda:0"
torch.manual_seed(0)
M, C = 8192, 128
for dtype in (torch.bfloat16, torch.float16):
    z = (torch.randn(M, C, device=dev) * 2 + 0.7).to(dtype)
    dy = torch.randn(M, C, device=dev).to(dtype)
    mean = torch.randn(C, device=dev) * 0.5
    rstd = 1.0 / (torch.rand(C, device=dev) + 0.5)
    gamma = torch.rand(C, device=dev) + 0.5
    beta = torch.randn(C, device=dev) * 0.3
    out = torch.empty_like(z)
    call("dm_bn_act_fwd", ptr(z), ptr(out), L.dt(dtype), M, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), L.ACT_GELU)
    zd = z.double()
    xh = (zd - mean.double()) * rstd.double()
    u = xh * gamma.double() + beta.double()
    cdf = 0.5 * (1 + torch.erf(u / 2 ** 0.5))
    y = u * cdf
    err = (out.double() - y).abs()
    ulp = (y.to(dtype).double() - y).abs()
    print(dtype, "fwd: max abs err", err.max().item(), " mean err / mean rounding err", (err.mean() / ulp.mean()).item(),
          " mis-rounded fraction", (out != y.to(dtype)).float().mean().item())
    nblk = L.colstat_blocks(M)
    p1, p2 = torch.empty(nblk, C, device=dev), torch.empty(nblk, C, device=dev)
    call("dm_bn_act_bwd_reduce", ptr(z), ptr(dy), L.dt(dtype), M, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), L.ACT_GELU, ptr(p1), ptr(p2))
    pdf = torch.exp(-0.5 * u * u) / (2 * torch.pi) ** 0.5
    g = dy.double() * (cdf + u * pdf)
    s1, s2 = g.sum(0), (g * xh).sum(0)
    print("   reduce: rel err s1", ((p1.double().sum(0) - s1).abs().max() / s1.abs().max()).item(), " s2", ((p2.double().sum(0) - s2).abs().max() / s2.abs().max()).item())
    dz = torch.empty_like(z)
    s1f, s2f = s1.float(), s2.float()
    call("dm_bn_act_bwd_apply", ptr(z), ptr(dy), ptr(dz), L.dt(dtype), M, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), L.ACT_GELU, ptr(s1f), ptr(s2f))
    ref = gamma.double() * rstd.double() * (g - s1 / M - xh * s2 / M)
    err = (dz.double() - ref).abs()
    ulp = (ref.to(dtype).double() - ref).abs()
    print("   apply: max abs err", err.max().item(), " mean err / mean rounding err", (err.mean() / ulp.mean()).item(), " mis-rounded fraction", (dz != ref.to(dtype)).float().mean().item())
