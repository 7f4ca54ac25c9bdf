"""Do launch-bound small kernels on one stream overlap with the full-chip halo weight-gradient kernel on another?
Times: (a) 6 wgrad launches alone, (b) 600 small launches alone, (c) both on one stream, (d) on two streams."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionmodel_amd import ops
from diffusionmodel_amd._lib import call, ptr, dt

dev = "cuda:0"
B, H, Ci, Co = 64, 64, 128, 128
x = torch.randn(B, H, H, Ci, device=dev).bfloat16()
y = torch.randn(B, H, H, Co, device=dev).bfloat16()
dw = torch.zeros(Co, 3, 3, Ci, device=dev)
db = torch.zeros(Co, device=dev)
geom = dict(dtype=torch.bfloat16, B=B, Hi=H, Wi=H, C1=Ci, C2=0, Hq=H, Wq=H, sy=1, sx=1, T=9, KW=3, ty=1, tx=1, oy0=-1, ox0=-1, Ho=H, Wo=H, N=Co)
part = torch.randn(60000, 4, device=dev)        # one workgroup walks 60000 rows: ~10-20 us of a single CU, like the strip / dense-layer launches
small_out = torch.empty(4, device=dev)


def wg():
    for _ in range(6):
        ops._wgrad_call(y, x, None, dw, db, ldy=Co, ldw=9 * Ci, **geom)


def small():
    for _ in range(200):
        call("dm_col_reduce", ptr(part), 60000, 4, ptr(small_out), 0)


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


s2 = torch.cuda.Stream()


def both_serial():
    wg()
    small()


def both_two_streams():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        wg()
    small()
    torch.cuda.current_stream().wait_stream(s2)


print(f"wgrad x6 alone      {timed(wg):7.3f} ms")
print(f"small x200 alone    {timed(small):7.3f} ms")
print(f"one stream          {timed(both_serial):7.3f} ms")
print(f"two streams         {timed(both_two_streams):7.3f} ms")
