"""Designed probe for the r02 corruption under GPU sharing: ONE kernel — the halo-resident 3x3 convolution
(156 KiB of LDS, LDS-DMA staging, MFMA) — on integer data, the same launch repeated `reps` times, every result compared with
the first (bit-exact when nothing interferes).  Run one copy (control) and two concurrent copies (DM_DEVICE_GUARD=0).

    python scripts/share_conv_probe.py <tag> <reps> [variant]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusionmodel_amd import _lib as L, ops as o

tag, reps = sys.argv[1], int(sys.argv[2])
if len(sys.argv) > 3:
    L.load().dm_set_conv_variant(int(sys.argv[3]))
dev = "cuda:0"
g = torch.Generator().manual_seed(3)
B, H, C, N = 64, 32, 256, 256
x = torch.randint(-3, 4, (B, H, H, C), generator=g).float().to(dev).bfloat16()
w = torch.nn.Parameter(torch.randint(-2, 3, (N, C, 3, 3), generator=g).float().to(dev).contiguous(memory_format=torch.channels_last))


class Hd:
    weight, bias = w, None


sp = o.ConvSpec(3, 3, 1, 1)
with torch.no_grad():
    y0 = o.conv_bn_act(x, None, Hd, None, sp)
    path = L.load().dm_last_conv_path()
    torch.cuda.synchronize()
    bad, first = 0, None
    t0 = time.time()
    for r in range(reps):
        y = o.conv_bn_act(x, None, Hd, None, sp)
        if r % 16 == 15 or r == reps - 1:
            pass
        ne = (y != y0)
        nb = int(ne.sum())
        if nb:
            bad += 1
            if first is None:
                idx = ne.nonzero()
                first = (r, nb, idx[0].tolist(), idx[-1].tolist(), sorted(set(idx[:, 0].tolist()))[:8], sorted(set((idx[:, 1] // 8).tolist()))[:8],
                         sorted(set((idx[:, 3] // 128).tolist())))
    torch.cuda.synchronize()
print(f"[{tag}] conv path {path} variant {L.load().dm_get_conv_variant()}: {bad} of {reps} repetitions differ from the first result; "
      f"first bad: {first}; {time.time() - t0:.1f} s", flush=True)
