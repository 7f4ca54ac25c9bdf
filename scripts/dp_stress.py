"""Diagnosis aid for the one-GPU two-rank rehearsal (scripts/dp_rehearsal.py): repeats the local backward pass and the reduced
backward pass and reports, per repetition, (a) how far a LOCAL gradient is from the first local one (kernel nondeterminism) and
(b) how far the REDUCED gradient is from the host-side sum — so that a mismatch can be attributed to the kernels or to the reducer.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29641 scripts/dp_stress.py [reps]
"""
import os
import sys

import torch
import torch.distributed as dist

# Two processes time-share the GPU here, and on this driver stack a workgroup that holds more than 64 KiB of LDS does not survive
# being preempted for the other process (scripts/race_hunt.py: 7 of 88 backward passes corrupted with the 156-KiB halo kernels,
# 0 of 60 with every kernel at <= 64 KiB; a process that has the GPU to itself: 0 of 40).  The rehearsal checks the reducer,
# not the kernels, so it runs the <= 64-KiB kernel variants; one process per GPU — the product configuration — is not affected.
os.environ.setdefault("DM_CONV_VARIANT", "2")
os.environ.setdefault("DM_WGRAD_VARIANT", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffusionmodel_amd as D
from diffusionmodel_amd import parallel

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
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


def run(reducer):
    torch.manual_seed(7 + rank)
    ddpm._rng_calls = 0
    ddpm._rng_dev = None
    opt.zero_grad()
    if reducer is not None:
        reducer.begin()
    loss = ddpm(x, c, am)
    loss.backward()
    if reducer is not None:
        reducer.finish()
    else:
        opt.gather_grads()
    torch.cuda.synchronize()
    return opt.flat_g.clone()


def worst(got, ref):
    bad = []
    for (p, off, n), name in zip(opt._slots, names):
        rn = ref[off:off + n].norm().item()
        if rn > 1e-9:
            d = (got[off:off + n] - ref[off:off + n]).norm().item() / rn
            if d > 1e-3:
                bad.append((name, round(d, 4)))
    return bad


local0 = run(None)
ref = local0.cpu()
dist.all_reduce(ref)
ref = ref.to(dev)
for r in range(reps):
    loc = run(None)
    dl = ((loc - local0).norm() / local0.norm()).item()
    got = run(red)
    dr = ((got - ref).norm() / ref.norm()).item()
    bl, br = worst(loc, local0), worst(got, ref)
    print(f"rank {rank} rep {r}: local vs first local {dl:.2e} ({len(bl)} params > 1e-3: {bl[:4]}); reduced vs host sum {dr:.2e} "
          f"({len(br)} params > 1e-3: {br[:4]})", flush=True)
dist.destroy_process_group()
