"""Rehearsal of the data-parallel step on ONE GPU: 2 ranks (torchrun --nproc-per-node 2), both on cuda:0, gloo backend
(RCCL refuses two ranks per device).  It exercises what the CPU tests cannot: OverlappedGradReducer driven by the real
backward pass (ON_WGRAD notifications from the HIP weight-gradient launches, side stream, packed `.grad` reduce), and
checks the reduced gradient of the observation step and of two overlapped steps against a blocking host-side sum of the
ranks' local gradients, parameter by parameter.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 scripts/dp_rehearsal.py
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffusionmodel_amd as D
from diffusionmodel_amd import parallel

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)

torch.manual_seed(0)
net = D.ContextUnet(3, 64, 4, bottleneck_k=4, dtype=torch.bfloat16)
ddpm = D.DDPM(net, (1e-4, 0.02), 1000, dev, drop_prob=0.1)
ddpm.train()
ddpm.rng_seed = 100 + rank
opt = D.FusedAdamW(ddpm.parameters(), lr=1e-4, grad_scale=1.0 / world)
names = [n for n, _ in ddpm.named_parameters()]
g = torch.Generator().manual_seed(50 + rank)
x = (torch.rand(8, 3, 64, 64, generator=g) * 2 - 1).to(dev)
c = torch.randint(0, 4, (8,), generator=g).to(dev)
am = torch.ones(8, 64, 64, device=dev)
red = parallel.OverlappedGradReducer(opt, n_buckets=5)


def backward_only(reducer):
    """One forward/backward with identical random draws every call; returns the flat gradient after reduction (or the local one)."""
    torch.manual_seed(7 + rank)
    ddpm._rng_calls = 0
    ddpm._rng_dev = None
    opt.zero_grad()
    if reducer is not None:
        reducer.begin()
    loss = ddpm(x, c, am)
    loss.backward()
    early = sum(reducer._launched) if reducer is not None else 0
    if reducer is not None:
        reducer.finish()
    else:
        opt.gather_grads()
    torch.cuda.synchronize()
    return opt.flat_g.clone(), float(loss), early


local, l0, _ = backward_only(None)
ref = local.cpu()
dist.all_reduce(ref)                                   # blocking host-side sum of the ranks' local gradients
ref = ref.to(dev)
worst = 0.0
for step in range(3):                                  # step 0 = observation (no early launches), 1-2 overlapped
    got, l1, early = backward_only(red)
    rel = ((got - ref).norm() / ref.norm()).item()
    bad = []
    for (p, off, n), name in zip(opt._slots, names):
        d = (got[off:off + n] - ref[off:off + n]).norm().item() / max(ref[off:off + n].norm().item(), 1e-30)
        if d > 1e-3 and ref[off:off + n].norm().item() > 1e-12:
            bad.append((name, d))
    print(f"rank {rank} step {step}: loss {l1:.5f} (local run {l0:.5f}), buckets launched during backward {early}/{len(red.buckets)}, "
          f"|reduced - host sum| / |host sum| = {rel:.2e}, parameters off by > 1e-3: {len(bad)} {bad[:3]}")
    worst = max(worst, rel)
    assert (early == 0) == (step == 0), early
assert worst < 1e-5, worst                            # only fp32 atomic ordering of the non-halo weight-gradient kernels differs
dist.destroy_process_group()
