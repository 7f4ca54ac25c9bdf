"""Capture the cfg-2 train step as a launch plan, print what is in it (foreign = torch-issued kernels) and time replay vs eager."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffusionmodel_amd as D
from bench import synthetic_batch

F = int(os.environ.get("F", "128")); B = int(os.environ.get("B", "64"))
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = D.ContextUnet(3, F, 4, bottleneck_k=4, dtype=torch.bfloat16)
ddpm = D.DDPM(net, (1e-4, 0.02), 1000, dev, drop_prob=0.1)
ddpm.train()
opt = D.FusedAdamW(ddpm.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0)
x, c, am = synthetic_batch(B, 64, 4, dev)
step = D.GraphedTrainStep(ddpm, opt, x, c, am, mode=os.environ.get("MODE", "plan"))
if step.plan is not None:
    p = step.plan
    print("ops", p.n_ops, "kernels", p.n_kernels, "memsets", p.n_memsets, "markers", p.n_markers, "skipped", p.n_skipped, "segments", p.n_segments)
    tally = {}
    for kind, nm in p.op_names():
        tally[nm] = tally.get(nm, 0) + 1
    for nm, n in sorted(tally.items(), key=lambda kv: -kv[1]):
        if "dm" not in nm or os.environ.get("ALL"):
            print(f"  x{n}: {nm[:160]}")
for _ in range(3):
    step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    loss = step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t = time.perf_counter() - t0
print(f"replay: host enqueue {t_enq / K * 1e3:.2f} ms/step, wall {t / K * 1e3:.2f} ms/step, loss {loss.item():.4f}")
# host cost of ONE step with an empty queue (no back-pressure from the device)
for name, fn in (("replay", step), ("eager", step._eager)):
    hs = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        hs.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    print(f"{name}: host time of one step on an idle queue: " + " ".join(f"{h * 1e3:.2f}" for h in hs) + " ms")
if step.plan is not None:
    ms = step.plan.run_timed("conv3x3_halo_kernel")
    print(f"halo launches {len(ms)}, total {sum(ms):.3f} ms, avg {sum(ms) / max(len(ms), 1) * 1e3:.1f} us")
