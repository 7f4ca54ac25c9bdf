"""Per-shape timing of the fp32 dense-layer kernels (dm_linear_fwd / dm_linear_bwd) on the cfg-2 shapes.  GPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionmodel_amd._lib import call, ptr

DEV = "cuda:0"
SHAPES = [("EmbedFC 1024x1024", 64, 1024, 1024), ("EmbedFC 512x512", 64, 512, 512), ("ctx 4->1024", 64, 4, 1024),
          ("SE 1024->64", 64, 1024, 64), ("SE 64->1024", 64, 64, 1024), ("SE 128->8", 64, 128, 8), ("SE 8->128", 64, 8, 128),
          ("CA1 conv1 128->8", 2048, 128, 8), ("CA1 proj 8->8", 2048, 8, 8), ("CA1 conv_h 8->128", 2048, 8, 128),
          ("CA2 256->16", 1024, 256, 16), ("CA2 16->256", 1024, 16, 256), ("CA3 512->32", 512, 512, 32), ("CA3 32->512", 512, 32, 512),
          ("CA4 1024->64", 256, 1024, 64), ("CA4 64->64", 256, 64, 64), ("CA4 64->1024", 256, 64, 1024)]


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for name, M, K, N in SHAPES:
    x, w, b = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV), torch.randn(N, device=DEV)
    y, g = torch.empty(M, N, device=DEV), torch.randn(M, N, device=DEV)
    dx, dw, db = torch.empty(M, K, device=DEV), torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
    t_f = timeit(lambda: call("dm_linear_fwd", ptr(x), ptr(w), ptr(b), ptr(y), M, K, N, 0))
    t_dx = timeit(lambda: call("dm_linear_bwd", ptr(x), ptr(w), ptr(g), ptr(dx), None, None, M, K, N))
    t_dw = timeit(lambda: call("dm_linear_bwd", ptr(x), ptr(w), ptr(g), None, ptr(dw), ptr(db), M, K, N))
    print(f"{name:22s} M={M:5d} K={K:5d} N={N:5d}  fwd {t_f:6.1f} us  dx {t_dx:6.1f} us  dw+db {t_dw:6.1f} us")
