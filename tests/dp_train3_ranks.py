"""TEST INFRASTRUCTURE (launched by tests/test_gpu_model.py::test_driver_engine_two_ranks_reproduce_the_reference_accumulation).

The drop-in training loop's engine (diffusionmodel_amd/train.py: what new_scripy.train_model drives) as TWO data-parallel
ranks against the reference's single-process trajectory with ACCUM_STEPS = 2 (tests/golden/train3.npz: new_scripy.py:777-803 run
by the real reference on the CPU — three optimiser steps, six micro-batches, injected draws).  Rank r computes micro-batches
r, r + 2, r + 4 (micro-batch m -> rank m % world); with world == ACCUM_STEPS the step must be the reference's step:
micro-batch losses, pre-clip gradient norms, parameter norms after step 3.

Both ranks share cuda:0 over gloo (RCCL refuses two ranks per device), so the library runs its <= 64-KiB-LDS kernels
(parallel.guard_shared_device); fp32.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29641 tests/dp_train3_ranks.py
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffusionmodel_amd as D
from diffusionmodel_amd import parallel
from diffusionmodel_amd.train import TrainEngine
from oracle import synth                       # (test infrastructure: the name-keyed synthetic weights / inputs of the fixtures)

os.environ["LOCAL_RANK"] = "0"
rank, world, local = parallel.init_from_env("gloo")
assert world == 2 and parallel.SHARED_DEVICE[0], "two ranks on one device expected (the guard must have engaged)"
assert D._lib.load().dm_get_conv_variant() == 2, "guard_shared_device must have selected the <= 64-KiB kernels"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)

g = np.load(os.path.join(ROOT, "tests", "golden", "train3.npz"))
schema = json.load(open(os.path.join(ROOT, "tests", "golden", "schema.json")))
lr, wd, accum, n_opt, B = [float(v) for v in g["hyper"]]
accum, n_opt, B = int(accum), int(n_opt), int(B)
assert accum == world
S, n_T = 64, 1000
ddpm = D.DDPM(D.ContextUnet(3, 32, 4, bottleneck_k=4, dtype=torch.float32), (1e-4, 0.02), n_T, dev, drop_prob=0.1)
sd = {k: synth.synth_tensor(k, tuple(s)) for k, s in schema["ddpm_fwd64"]}
for k in D.SCHEDULE_KEYS:
    sd[k] = D.ddpm_schedules(1e-4, 0.02, n_T)[k]
ddpm.load_state_dict(sd)
ddpm.train()
opt = D.FusedAdamW(ddpm.parameters(), lr=lr, weight_decay=wd, max_grad_norm=1.0)
eng = TrainEngine(ddpm, opt, accum_steps=accum, n_buckets=3, use_plan=False)
assert (eng.G, eng.local_accum) == (2, 1) and opt.grad_scale == 0.5

losses, norms = {}, []
for step in range(n_opt):
    m = step * world + rank
    tag = f"train3.m{m}"
    x = synth.synth_input(tag + ".x", (B, 3, S, S)).to(dev)
    c = torch.tensor([(m + i) % 4 for i in range(B)]).to(dev)
    am = synth.synth_attn_mask(B, S).to(dev)
    noise = synth.synth_noise(tag + ".noise", (B, 3, S, S)).to(dev)
    loss = eng.micro_batch(x, c, am, ts=torch.tensor(g[f"ts.{m}"]).to(dev), noise=noise, ctx_mask=torch.tensor(g[f"keep.{m}"]).to(dev))
    losses[m] = float(loss) * eng.loss_div
    norms.append(float(opt.grad_norm()))
    early = sum(1 for b in eng.reducer._order) if step else 0
allv = [None] * world
dist.all_gather_object(allv, losses)
losses = {**allv[0], **allv[1]}
ref_l, ref_n = [float(v) for v in g["losses"]], [float(v) for v in g["grad_norms"]]
l_err = [abs(losses[m] - ref_l[m]) / abs(ref_l[m]) for m in range(accum * n_opt)]
n_err = [abs(a - b) / abs(b) for a, b in zip(norms, ref_n)]
pn_err = {}
for cn, ch in ddpm.nn_model.named_children():
    ps = [p.detach().double().reshape(-1) for p in ch.parameters()]
    if ps and f"pnorm.{cn}" in g.files:
        pn_err[cn] = abs(float(torch.cat(ps).norm().item() / float(g[f"pnorm.{cn}"]) - 1.0))
# both ranks hold the same weights after the three steps
mine = opt.flat_p.cpu()
ref0 = mine.clone()
dist.broadcast(ref0, src=0)
spread = float((mine - ref0).abs().max())
print(f"rank {rank} dp_train3: losses {[round(losses[m], 6) for m in sorted(losses)]} ref {[round(v, 6) for v in ref_l]} "
      f"max loss rel err {max(l_err):.2e}, grad-norm rel err {[f'{v:.1e}' for v in n_err]}, worst child parameter-norm err "
      f"{max(pn_err.values()):.1e}, rank spread {spread:.1e}, optimiser steps {eng.opt_steps}", flush=True)
# the bars of test_three_optimiser_steps_reproduce_the_reference_loop_fp32 (single process, same fixture)
assert max(l_err) <= 2e-4, l_err
assert max(n_err) <= 5e-3, n_err
assert max(pn_err.values()) <= 2e-5, pn_err
assert spread <= 1e-6, spread
assert eng.opt_steps == n_opt and opt._step == n_opt
dist.barrier()
if rank == 0:
    print("dp_train3 OK", flush=True)
dist.destroy_process_group()
