"""Time the weight gradient of the cfg-2 1x1 layers: wgrad_pw_kernel vs the per-tap kernel, per shape (rotating buffers so the
operands come from HBM, not the L2 / MALL)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionmodel_amd import ops as o, _lib

lib = _lib.load()
dev = "cuda:0"
SHAPES = [(64, 32, 128), (64, 128, 32), (32, 32, 256), (32, 128, 32), (16, 64, 512), (16, 256, 64), (8, 128, 1024), (8, 512, 128)]
B = 64
NBUF = 6
targets = [int(t) for t in os.environ.get("TARGETS", "0,768").split(",")]
for H, C, N in SHAPES:
    M = B * H * H
    xs = [torch.randn(B, H, H, C, device=dev).bfloat16() for _ in range(NBUF)]
    dys = [torch.randn(B, H, H, N, device=dev).bfloat16() for _ in range(NBUF)]
    dw = torch.zeros(N, C, device=dev)
    db = torch.zeros(N, device=dev)
    line = f"{H:2d}^2 C{C:4d} N{N:4d}  ideal {(M * (C + N) * 2) / 5e12 * 1e6:6.1f} us @5TB/s |"
    for tgt in targets:
        lib.dm_set_wgrad_pw(1 if tgt else 0, tgt, 0)
        def run(i):
            o._wgrad_call(dys[i % NBUF], xs[i % NBUF], None, dw, db, dtype=torch.bfloat16, B=B, Hi=H, Wi=H, C1=C, C2=0, Hq=H, Wq=H, sy=1, sx=1, T=1, KW=1,
                          ty=1, tx=1, oy0=0, ox0=0, Ho=H, Wo=H, N=N, ldy=N, ldw=C)
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = 30
        e0.record()
        for i in range(K):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        line += f"  tgt {tgt:5d}: {e0.elapsed_time(e1) / K * 1e3:7.1f} us"
    print(line, flush=True)
