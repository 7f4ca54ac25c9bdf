"""Time dm_sumsq + dm_adamw on a flat buffer of the cfg-2 parameter count (GPU).  DM_ADAMW_BLOCKS overrides the grid."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionmodel_amd._lib import call, ptr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 43_600_000
dev = "cuda:0"
p, g = torch.randn(n, device=dev), torch.randn(n, device=dev) * 1e-3
m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
p16 = torch.zeros(n, device=dev, dtype=torch.bfloat16)
ss = torch.ones(1, device=dev)
hy = torch.tensor([1e-4, 0.9, 0.999, 1e-8, 1e-5, 1.0, 1.0, 0.1, 0.001], device=dev)
for shadow in (None, p16):
    for _ in range(3):
        call("dm_adamw", ptr(p), ptr(g), ptr(m), ptr(v), n, ptr(ss), ptr(hy), ptr(shadow))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call("dm_adamw", ptr(p), ptr(g), ptr(m), ptr(v), n, ptr(ss), ptr(hy), ptr(shadow))
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    by = n * (28 + (2 if shadow is not None else 0))
    print(f"adamw n={n} shadow={shadow is not None}: {t * 1e6:.1f} us  {by / t / 1e12:.2f} TB/s")
