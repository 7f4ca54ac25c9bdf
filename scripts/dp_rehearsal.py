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

# Two processes time-share the GPU here, and on this driver stack a workgroup that holds more than 64 KiB of LDS does not survive
# being preempted for the other process (scripts/race_hunt.py: 7 of 88 backward passes corrupted with the 156-KiB halo kernels,
# 0 of 60 with every kernel at <= 64 KiB; a process that has the GPU to itself: 0 of 40).  The rehearsal checks the reducer,
# not the kernels, so it runs the <= 64-KiB kernel variants; one process per GPU — the product configuration — is not affected.
os.environ.setdefault("DM_CONV_VARIANT", "2")
os.environ.setdefault("DM_WGRAD_VARIANT", "2")
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

# ---- the planned data-parallel step (bench.py's default): the captured step split into segments at the reducer's markers, the
# all-reduces between them — against the eager data-parallel step from the same state, same draws
def eager_step():
    opt.zero_grad()
    red.begin()
    loss = ddpm(x, c, am)
    loss.backward()
    red.finish()
    opt.step()
    return loss


def body(st):
    opt.zero_grad()
    red.begin(capture=torch.cuda.is_current_stream_capturing())
    loss = ddpm(st.x, st.c, st.am)
    loss.backward()
    red.finish()
    opt.step()
    return loss


def reset_draws():
    """Timesteps, keep masks and noise come from the library's Philox stream at a device-resident offset: rewind it in place (the
    captured step holds a pointer to this very tensor)."""
    ddpm._rng_calls = 0
    if getattr(ddpm, "_rng_dev", None) is not None:
        ddpm._rng_dev.zero_()


snap = dict(p=opt.flat_p.clone(), m=opt.exp_avg.clone(), v=opt.exp_avg_sq.clone(), t=opt._step_dev.clone(), step=opt._step,
            bufs=[b.clone() for b in ddpm.buffers()])


def restore():
    with torch.no_grad():
        opt.flat_p.copy_(snap["p"]); opt.exp_avg.copy_(snap["m"]); opt.exp_avg_sq.copy_(snap["v"]); opt._step_dev.copy_(snap["t"])
        for b, s0 in zip(ddpm.buffers(), snap["bufs"]):
            b.copy_(s0)
    opt._step = snap["step"]
    opt.refresh_shadow()
    from diffusionmodel_amd import ops
    ops.bump_weight_epoch()
    ops.refresh_packs()


def rank_spread(tag):
    """max |p_rank - p_rank0| over the flat parameter buffer (0 when the ranks hold identical weights) and the worst parameter."""
    mine = opt.flat_p.cpu()
    ref0 = mine.clone()
    dist.broadcast(ref0, src=0)
    diff = (mine - ref0).abs()
    worst = ""
    if diff.max() > 0:
        for (pp, off, n), name in zip(opt._slots, names):
            if diff[off:off + n].max() > 0:
                worst = f"{name} {diff[off:off + n].max().item():.2e}"
                break
    d = torch.tensor([diff.max().item()])
    dist.all_reduce(d, op=dist.ReduceOp.MAX)
    print(f"rank {rank} {tag}: max |p - p_rank0| = {d.item():.3e} {worst}")
    return d.item()


reset_draws()
assert rank_spread("before the optimiser steps") == 0.0
eager_losses = [float(eager_step()) for _ in range(2)]
# (gloo reduces device tensors through the host; its two ranks may differ in the last bit of a sum, RCCL's ring does not: allow ulps)
assert rank_spread("after 2 eager data-parallel steps") <= 1e-6
p_eager = opt.flat_p.clone()
restore()
planned = D.GraphedTrainStep(ddpm, opt, x, c, am, mode="plan", body=body, runner=red.replay)
restore()                                             # (the constructor restores too; the draws below must start where the eager run started)
reset_draws()
plan_losses = [float(planned()) for _ in range(2)]
torch.cuda.synchronize()
rel = ((opt.flat_p - p_eager).norm() / (p_eager - snap["p"]).norm().clamp_min(1e-30)).item()
print(f"rank {rank} planned DP: {planned.plan.n_kernels} kernels in {planned.plan.n_segments} segments (markers {planned.plan.segment_markers}), "
      f"losses eager {eager_losses} planned {plan_losses}, |p_plan - p_eager| / |update| = {rel:.2e}")
assert planned.plan.n_segments >= 3 and red.MARK_BACKWARD_DONE in planned.plan.segment_markers
for a, b in zip(eager_losses, plan_losses):
    assert abs(a - b) <= 2e-3 * abs(a), (eager_losses, plan_losses)
assert rel < 0.3, rel                                # Adam turns rounding-level gradient noise into +-lr moves: same band as the single-GPU graph test
assert rank_spread("after 2 planned data-parallel steps") <= 1e-6, "the ranks' parameters diverged after the planned steps"
dist.destroy_process_group()
