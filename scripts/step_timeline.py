"""Per-launch timeline of the planned cfg-2 train step: every op of the plan with its own HIP-event duration (mean of 3 instrumented
replays), in issue order, plus a per-kernel tally split into launches under / over 8 us.  Writes gpurun_out/step_timeline.txt."""
import os, sys, re, collections, torch
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
for _ in range(5):
    step()
plan = step.plan
acc = collections.defaultdict(list)
for _ in range(3):
    plan.run_timed("")
    for op, nm, ms in plan.timed_results(cap=16384):
        acc[op].append((nm, ms))
def short(nm):
    nm = re.sub(r"^_Z[A-Z]*\d+", "", nm)
    return nm[:70]
rows = [(op, short(v[0][0]), sum(m for _, m in v) / len(v) * 1e3) for op, v in sorted(acc.items())]
out = []
tot = sum(r[2] for r in rows)
out.append(f"{len(rows)} timed launches, sum of durations {tot/1e3:.3f} ms")
tally = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
for op, nm, us in rows:
    t = tally[nm]
    if us < 8: t[0] += 1; t[1] += us
    else: t[2] += 1; t[3] += us
out.append(f"{'kernel':72s} {'n<8us':>6s} {'us':>8s} {'n>=8us':>6s} {'us':>9s}")
for nm, t in sorted(tally.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
    out.append(f"{nm:72s} {t[0]:6d} {t[1]:8.1f} {t[2]:6d} {t[3]:9.1f}")
small = sum(t[0] for t in tally.values()); small_us = sum(t[1] for t in tally.values())
out.append(f"launches under 8 us: {small} of {len(rows)}, {small_us/1e3:.3f} ms")
out.append("")
for op, nm, us in rows:
    out.append(f"{op:5d} {us:8.1f}  {nm}")
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/step_timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:60]))
