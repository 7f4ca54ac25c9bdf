"""Find the backward op whose result differs between repetitions of the SAME train-step backward pass (GPU shared by several
processes: start 2+ copies with torchrun).  Every autograd.Function.backward of the operator layer is wrapped to checksum what
it returns; repetitions are compared with the first one in backward order.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29651 scripts/race_hunt.py [reps]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffusionmodel_amd as D
from diffusionmodel_amd import ops, modules

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rank = int(os.environ.get("RANK", "0"))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
LOG = []


def wrap(cls):
    orig = cls.backward

    def bw(ctx, *g):
        out = orig(ctx, *g)
        outs = out if isinstance(out, tuple) else (out,)
        for k, o in enumerate(outs):
            if isinstance(o, torch.Tensor) and o.is_floating_point() and o.numel() > 0:
                LOG.append((f"{cls.__name__}[{k}]{tuple(o.shape)}", o.detach().double().abs().sum()))
        return out
    cls.backward = staticmethod(bw)


for mod in (ops, modules):
    for name, obj in list(vars(mod).items()):
        if isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function:
            wrap(obj)

torch.manual_seed(0)
net = D.ContextUnet(3, 64, 4, bottleneck_k=4, dtype=torch.bfloat16)
ddpm = D.DDPM(net, (1e-4, 0.02), 1000, dev, drop_prob=0.1)
ddpm.train()
ddpm.rng_seed = 100 + rank
opt = D.FusedAdamW(ddpm.parameters(), lr=1e-4)
names = [n for n, _ in ddpm.named_parameters()]
g = torch.Generator().manual_seed(50 + rank)
x = (torch.rand(8, 3, 64, 64, generator=g) * 2 - 1).to(dev)
c = torch.randint(0, 4, (8,), generator=g).to(dev)
am = torch.ones(8, 64, 64, device=dev)


def run():
    LOG.clear()
    torch.manual_seed(7 + rank)
    ddpm._rng_calls = 0
    ddpm._rng_dev = None
    opt.zero_grad()
    loss = ddpm(x, c, am)
    loss.backward()
    opt.gather_grads()
    torch.cuda.synchronize()
    return [n for n, _ in LOG], torch.stack([v for _, v in LOG]).cpu(), opt.flat_g.clone()


n0, s0, f0 = run()
for r in range(reps):
    n1, s1, f1 = run()
    assert n1 == n0
    rel = ((s1 - s0).abs() / s0.abs().clamp_min(1e-30))
    bad = (rel > 1e-4).nonzero().flatten().tolist()
    dflat = ((f1 - f0).norm() / f0.norm()).item()
    if bad:
        first = bad[0]
        ctx = ", ".join(f"#{i} {n0[i]} {rel[i]:.1e}" for i in bad[:6])
        prev = f"#{first - 1} {n0[first - 1]} {rel[first - 1]:.1e}" if first else "-"
        print(f"rank {rank} rep {r}: flat grad off by {dflat:.2e}; {len(bad)}/{len(n0)} backward results differ; first: {ctx}; before it: {prev}", flush=True)
    else:
        print(f"rank {rank} rep {r}: flat grad off by {dflat:.2e}; all {len(n0)} backward results repeat", flush=True)
