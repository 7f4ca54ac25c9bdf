"""What does ONE cross-stream event round trip per step cost, and does the side stream's priority matter?  (DESIGN.md section 6:
the data-parallel step pays ~0.3 ms for the hand-off to the RCCL stream even when the collective is free.)
Replays the planned cfg-2 train step 40 times per variant."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffusionmodel_amd as D
from bench import synthetic_batch
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = D.ContextUnet(3, 128, 4, bottleneck_k=4, dtype=torch.bfloat16)
ddpm = D.DDPM(net, (1e-4, 0.02), 1000, dev, drop_prob=0.1).train()
opt = D.FusedAdamW(ddpm.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0)
x, c, am = synthetic_batch(64, 64, 4, dev)
step = D.GraphedTrainStep(ddpm, opt, x, c, am, mode="plan")
plan = step.plan
n_ops = plan.n_ops


def run(kind, side, steps=40):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if kind == "none":
            step()
        elif kind == "pair_before":                     # one round trip, before the step
            side.wait_stream(torch.cuda.current_stream()); torch.cuda.current_stream().wait_stream(side)
            step()
        elif kind == "record_only":                     # the side stream waits on us, we never wait on it
            side.wait_stream(torch.cuda.current_stream())
            step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


import ctypes
hip = ctypes.CDLL("libamdhip64.so")
def raw_event(flags):
    e = ctypes.c_void_p()
    assert hip.hipEventCreateWithFlags(ctypes.byref(e), ctypes.c_uint(flags)) == 0
    return e
def raw_run(kind, side, ev_a, ev_b, steps=40):
    main = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream); sd = ctypes.c_void_p(side.cuda_stream)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        hip.hipEventRecord(ev_a, main)
        if kind != "record_nowait":
            hip.hipStreamWaitEvent(sd, ev_a, 0)
        if kind == "pair":
            hip.hipEventRecord(ev_b, sd); hip.hipStreamWaitEvent(main, ev_b, 0)
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
side0 = torch.cuda.Stream()
for fname, flags in (("disable_timing", 0x2), ("disable_timing|release_to_device", 0x2 | 0x40000000), ("disable_timing|no_system_fence", 0x2 | 0x20000000), ("default(timing)", 0x0)):
    for kind in ("record_nowait", "record_only", "pair"):
        print(f"raw {fname:34s} {kind:12s} {raw_run(kind, side0, raw_event(flags), raw_event(flags)):7.3f} ms/step", flush=True)
print(f"none {run('none', side0):7.3f}", flush=True)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
for name, side in (("default priority", torch.cuda.Stream()),):
    for kind in ("none", "pair_before", "record_only", "none"):
        print(f"{name:18s} {kind:12s} {run(kind, side):7.3f} ms/step", flush=True)
