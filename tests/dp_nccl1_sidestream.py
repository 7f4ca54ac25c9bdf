"""TEST INFRASTRUCTURE (launched by tests/test_gpu_model.py::test_dp_product_configuration_one_process_nccl_side_stream_load).

The data-parallel PRODUCT configuration on the one GPU a test box has: ONE process, `nccl` (= RCCL) with world size 1, the default
kernels (156 KiB of LDS per workgroup in the halo kernels), parallel.OverlappedGradReducer in eager and in launch-plan mode — and a
REAL load on the reducer's side stream while the backward pass runs: where a bucket's all-reduce is launched, a streaming kernel
(dm_cast over a 256-MB scratch buffer, `SIDE_PASSES` times) is launched behind it on the same side stream, standing in for the RCCL
ring kernels of an 8-GPU job (with one rank RCCL's all-reduce moves nothing).  That is the regime tests with gloo ranks cannot
reach: LDS-heavy workgroups on the launch stream WHILE another queue of the same process has work on the chip.

Checked: the flat gradient after finish() equals the no-reducer run of the same seeded step — EXACTLY on integer-valued data
(a conv + BatchNorm-free stack would be needed for that; the network's BatchNorm makes values non-integer), so here: within the
fp32-atomics band measured between two no-reducer runs of the same pass — over several repetitions, eager and planned.

    python tests/dp_nccl1_sidestream.py
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("LOCAL_RANK", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29800 + os.getpid() % 150))
import diffusionmodel_amd as D
from diffusionmodel_amd import _lib as L, parallel

SIDE_PASSES = int(os.environ.get("DM_SIDE_PASSES", "6"))
REPS = int(os.environ.get("DM_SIDE_REPS", "6"))

rank, world, local = parallel.init_from_env("nccl", force=True)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda", 0)
lib = L.load()
assert lib.dm_get_conv_variant() == L.DEFAULT_CONV_VARIANT, "the product configuration runs the default (halo-resident) kernels"

torch.manual_seed(0)
F_, B = 128, 16                                        # the benchmark width: every 3x3 layer takes the halo-resident kernels
net = D.ContextUnet(3, F_, 4, bottleneck_k=4, dtype=torch.bfloat16)
ddpm = D.DDPM(net, (1e-4, 0.02), 1000, dev, drop_prob=0.1)
ddpm.train()
ddpm.rng_seed = 4321
opt = D.FusedAdamW(ddpm.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, grad_scale=1.0)
g = torch.Generator().manual_seed(9)
x = (torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).to(dev)
c = torch.randint(0, 4, (B,), generator=g).to(dev)
am = torch.ones(B, 64, 64, device=dev)
am[:, 32:, :] = 3.0

red = parallel.OverlappedGradReducer(opt, n_buckets=6)
scratch_a = torch.empty(64 << 20, dtype=torch.float32, device=dev)          # 256 MB each
scratch_b = torch.empty(64 << 20, dtype=torch.float32, device=dev)
scratch_a.normal_()
side_launches = [0]
_plain_launch = red._launch


load_stream = red._stream if red._stream is not None else torch.cuda.Stream()


def _loaded_launch(b):
    """The reducer's bucket launch + a streaming load behind it on a second queue of this process — the reducer's own side stream
    when it has one, else (RCCL: the collectives run on the process group's stream) a stream of the test that is ordered after the
    launch stream exactly like a bucket's all-reduce (only when a collective really starts: not while a capture records markers)."""
    _plain_launch(b)
    if getattr(red, "_capture", False):
        return
    load_stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(load_stream):
        for _ in range(SIDE_PASSES):
            L.call("dm_cast", L.ptr(scratch_a), L.ptr(scratch_b), L.DM_F32, L.DM_F32, scratch_a.numel())
        side_launches[0] += SIDE_PASSES


red._launch = _loaded_launch


def reset_draws():
    ddpm._rng_calls = 0
    if getattr(ddpm, "_rng_dev", None) is not None:
        ddpm._rng_dev.zero_()


def backward(reducer):
    reset_draws()
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


def rel(a, b):
    return float((a - b).norm() / b.norm())


ref, l_ref, _ = backward(None)
ref2, _, _ = backward(None)
band = rel(ref2, ref)                                  # run-to-run spread of the pass itself (fp32 atomics of the non-halo weight gradients)
print(f"no-reducer run vs itself: {band:.2e} (loss {l_ref:.6f})", flush=True)
assert band < 1e-5, band
names = [n for n, _ in ddpm.named_parameters()]
worst = 0.0
for rep in range(REPS):
    got, l1, early = backward(red)
    d = rel(got, ref)
    bad = []
    for (p, off, n), name in zip(opt._slots, names):
        rn = ref[off:off + n].norm().item()
        if rn > 1e-12 and (got[off:off + n] - ref[off:off + n]).norm().item() / rn > 1e-3:
            bad.append(name)
    print(f"eager rep {rep}: buckets launched during backward {early}/{len(red.buckets)}, side-stream kernels so far {side_launches[0]}, "
          f"|g - g_ref| / |g_ref| = {d:.2e}, loss diff {abs(l1 - l_ref):.1e}, parameters off by > 1e-3: {bad[:4]}", flush=True)
    assert (early > 0) == (rep > 0), (rep, early)
    assert d <= max(10 * band, 2e-6) and not bad, (rep, d, bad[:8])
    worst = max(worst, d)
assert side_launches[0] >= SIDE_PASSES * len(red.buckets) * (REPS - 1)
assert scratch_b[:1024].equal(scratch_a[:1024]) and scratch_b[-1024:].equal(scratch_a[-1024:])       # the side load really ran


# ---- launch-plan mode: two planned data-parallel steps vs two plain (no reducer) eager steps from the same state and draws
def body(st):
    opt.zero_grad()
    red.begin(capture=torch.cuda.is_current_stream_capturing())
    loss = ddpm(st.x, st.c, st.am)
    loss.backward()
    red.finish()
    opt.step()
    return loss


snap = dict(p=opt.flat_p.clone(), m=opt.exp_avg.clone(), v=opt.exp_avg_sq.clone(), t=opt._step_dev.clone(), step=opt._step,
            bufs=[b.clone() for b in ddpm.buffers()])


def restore():
    from diffusionmodel_amd import ops
    with torch.no_grad():
        opt.flat_p.copy_(snap["p"]); opt.exp_avg.copy_(snap["m"]); opt.exp_avg_sq.copy_(snap["v"]); opt._step_dev.copy_(snap["t"])
        for b, s0 in zip(ddpm.buffers(), snap["bufs"]):
            b.copy_(s0)
    opt._step = snap["step"]
    opt.refresh_shadow()
    ops.bump_weight_epoch()
    ops.refresh_packs()


def plain_step():
    opt.zero_grad()
    loss = ddpm(x, c, am)
    loss.backward()
    opt.step()
    return float(loss)


reset_draws()
plain_losses = [plain_step()]
torch.cuda.synchronize()
g1_plain = opt.flat_g.clone()                          # the gradient of step 1 (same weights, same draws in both runs)
plain_losses.append(plain_step())
torch.cuda.synchronize()
p_plain = opt.flat_p.clone()
restore()
before = side_launches[0]
planned = D.GraphedTrainStep(ddpm, opt, x, c, am, mode="plan", body=body, runner=red.replay)
restore()
reset_draws()
plan_losses = [float(planned())]
torch.cuda.synchronize()
rel_g = rel(opt.flat_g, g1_plain)
plan_losses.append(float(planned()))
torch.cuda.synchronize()
upd = (p_plain - snap["p"]).norm().clamp_min(1e-30)
rel_p = float((opt.flat_p - p_plain).norm() / upd)
print(f"planned DP (nccl, 1 rank, side load): {planned.plan.n_kernels} kernels in {planned.plan.n_segments} segments, markers "
      f"{planned.plan.segment_markers}; losses plain {plain_losses} planned {plan_losses}; step-1 gradient vs the no-reducer eager step "
      f"{rel_g:.2e}; |p_plan - p_plain| / |update| after 2 steps = {rel_p:.2e}; side-stream kernels during the replays {side_launches[0] - before}", flush=True)
assert planned.plan.n_segments >= 3
assert side_launches[0] - before >= 2 * SIDE_PASSES * len(red.buckets)
assert rel_g <= max(10 * band, 2e-6), rel_g            # the reduced gradient of the planned step == the plain step's, to the atomics band
assert abs(plan_losses[0] - plain_losses[0]) <= 1e-6 * abs(plain_losses[0]) + 1e-7, (plan_losses, plain_losses)   # same weights, same draws
assert abs(plan_losses[1] - plain_losses[1]) <= 2e-3 * abs(plain_losses[1]), (plan_losses, plain_losses)
assert rel_p < 0.3, rel_p                              # Adam turns rounding-level gradient noise into +-lr moves (same band as the graph == eager test)
print("dp_nccl1_sidestream OK", flush=True)
dist.destroy_process_group()
