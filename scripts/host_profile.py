"""cProfile of the host side of the eager train step (where do the ~15 ms of Python per step go?).  GPU only."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffusionmodel_amd as D

dev = "cuda:0"
torch.manual_seed(0)
net = D.ContextUnet(3, 128, 4, bottleneck_k=4, dtype=torch.bfloat16)
ddpm = D.DDPM(net, (1e-4, 0.02), 1000, dev, drop_prob=0.1)
ddpm.train()
opt = D.FusedAdamW(ddpm.parameters(), lr=1e-4)
x = torch.rand(64, 3, 64, 64, device=dev)
c = torch.randint(0, 4, (64,), device=dev)
am = torch.ones(64, 64, 64, device=dev)


def step():
    opt.zero_grad()
    loss = ddpm(x, c, am)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)      # backward on this thread, so that cProfile sees the Python backward functions
step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
