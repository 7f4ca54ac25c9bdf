"""bn_act_fwd / bn_bwd_apply: the slot-folding kernels against the plain ones (what the per-workgroup fold prologue costs)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionmodel_amd import _lib as L
from diffusionmodel_amd.ops import call, ptr

dev = "cuda:0"
def timeit(fn, iters=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
d = L.dt(torch.bfloat16)
for (H, C) in [(64, 128), (32, 256), (16, 512), (8, 1024), (64, 32)]:
    M = 64 * H * H
    z = torch.randn(M, C, device=dev).bfloat16(); dy = torch.randn(M, C, device=dev).bfloat16(); out = torch.empty_like(z)
    mean, rstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    S = 8
    ps = torch.zeros(S, C, device=dev, dtype=torch.float64); pq = torch.full((S, C), float(M) / S, device=dev, dtype=torch.float64)
    s1, s2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    db, dg = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    t0 = timeit(lambda: call("dm_bn_act_fwd", ptr(z), ptr(out), d, M, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), L.ACT_GELU))
    t1 = timeit(lambda: call("dm_bn_act_fwd_slots", ptr(z), ptr(out), d, M, C, ptr(ps), ptr(pq), S, 1e-5, 0.1, ptr(gamma), ptr(beta), L.ACT_GELU, ptr(mean), ptr(rstd), ptr(rm), ptr(rv)))
    t2 = timeit(lambda: call("dm_bn_act_bwd_apply", ptr(z), ptr(dy), ptr(out), d, M, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), L.ACT_GELU, ptr(s1), ptr(s2)))
    t3 = timeit(lambda: call("dm_bn_act_bwd_apply_slots", ptr(z), ptr(dy), ptr(out), d, M, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), L.ACT_GELU, ptr(ps), ptr(ps), S, ptr(db), ptr(dg)))
    print(f"[64x{H}x{H}x{C}] fwd {t0:6.1f} us  fwd_slots {t1:6.1f} us | bwd_apply {t2:6.1f} us  bwd_apply_slots {t3:6.1f} us")
