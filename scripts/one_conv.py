"""Run the main cfg-2 conv shape (fwd + wgrad) a few times — target for rocprofv3 --pmc passes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusionmodel_amd import ops
dev, dtype = "cuda:0", torch.bfloat16
B, H, Ci, Co, k = 64, 64, 128, 128, 3
x = torch.randn(B, H, H, Ci, device=dev).to(dtype)
w = (torch.randn(Co, k, k, Ci, device=dev) / 34).to(dtype)
y = torch.empty(B, H, H, Co, device=dev, dtype=dtype)
geom = dict(dtype=dtype, B=B, Hi=H, Wi=H, C1=Ci, C2=0, Hq=H, Wq=H, sy=1, sx=1, T=9, KW=3, ty=1, tx=1, oy0=-1, ox0=-1, Ho=H, Wo=H, N=Co)
dw = torch.zeros(Co, k, k, Ci, device=dev)
for _ in range(4):
    ops._conv_call(x, None, w.data_ptr(), 9 * Ci, y, **geom)
    ops._wgrad_call(y, x, None, dw, None, ldy=Co, ldw=9 * Ci, **geom)
torch.cuda.synchronize()
