"""Which Python call sites issue device-to-device copies in the eager train step?  (torch.profiler with stacks; GPU only.)"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffusionmodel_amd as D

dev = "cuda:0"
torch.manual_seed(0)
net = D.ContextUnet(3, 128, 4, bottleneck_k=4, dtype=torch.bfloat16)
ddpm = D.DDPM(net, (1e-4, 0.02), 1000, dev, drop_prob=0.1)
ddpm.train()
opt = D.FusedAdamW(ddpm.parameters(), lr=1e-4)
x = torch.rand(16, 3, 64, 64, device=dev)
c = torch.randint(0, 4, (16,), device=dev)
am = torch.ones(16, 64, 64, device=dev)


def step():
    opt.zero_grad()
    loss = ddpm(x, c, am)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::add", "aten::add_", "aten::zeros", "aten::fill_", "aten::zero_"):
        st = [f for f in (ev.stack or []) if "diffusionmodel_amd" in f or "scripts" in f]
        cnt[(ev.name, st[0] if st else (ev.stack[0] if ev.stack else "?"), str(ev.input_shapes)[:60])] += 1
for (name, where, shp), n in cnt.most_common(40):
    print(n, name, where, shp)
