#!/usr/bin/env python3
"""Benchmark of the DDPM / ContextUnet hot path on MI355X (BASELINE.json metric: denoiser-steps/sec).

    python bench.py --gpus N --steps K --warmup W

Workload (N=1): BASELINE.json configs[1] — new_scripy.py ContextUnet 64x64, n_feat=128, T=1000, bf16,
batch 64 per GPU; a "step" is one full training step: DDPM.forward (q-sample + denoiser + weighted loss)
+ backward + clip_grad_norm_(1.0) + AdamW, on synthetic inputs already resident in HBM.
N>1: one process per GPU — under torchrun, or started by this script itself when WORLD_SIZE is not set (`python bench.py
--gpus N`: the parent makes no GPU call, starts N ranks and relays rank 0's line) — batch 64 per rank (weak scaling), RCCL
all-reduce of the flat gradient in buckets between the segments of the replayed step; `value` is the whole-job rate N*K/T.
Rank 0 prints ONE JSON line (`ranks` = the world size RCCL saw, `devices` = every rank's device).  The step runs as a launch
plan by default (`--exec plan|graph|eager`).

Extra objects in the line: `roofline` for the dominant kernel (the bf16 implicit-GEMM convolution:
algorithmic FLOPs of every launch inside the timed region / HIP-event time of those launches, against the
2.5 PFLOP/s dense bf16 MFMA peak), `cpu_baseline` (the CPU oracle — a port — timed on the host cores on a
bounded sample of the same workload) and `sample` (CFG sampling steps/s, n=64 -> denoiser batch 128).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


def synthetic_batch(B, S, n_classes, device, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, S, S, generator=g).clamp_(-1, 1)
    c = torch.randint(0, n_classes, (B,), generator=g)
    am = torch.full((B, S, S), 0.5)
    am[:, S // 2:, :] = 1.0
    for i in range(B):
        y0, x0 = int(torch.randint(0, S // 2, (1,), generator=g)), int(torch.randint(0, S // 2, (1,), generator=g))
        am[i, y0:y0 + S // 4, x0:x0 + S // 4] = 3.0
    return x.to(device), c.to(device), am.to(device)


def _rand_state(spec, prefix="nn_model."):
    """torch-default-like random parameters for an oracle spec (enough for timing)."""
    P = {}
    for k, shp in spec.items():
        if k.endswith("num_batches_tracked"):
            P[prefix + k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_var") or (k.endswith("weight") and len(shp) == 1):
            P[prefix + k] = torch.ones(shp)
        elif len(shp) >= 2:
            fan = 1
            for d in shp[1:]:
                fan *= d
            P[prefix + k] = (torch.rand(shp) * 2 - 1) / fan ** 0.5
        else:
            P[prefix + k] = torch.zeros(shp)
    return P


def _cpu_info():
    model, phys = "unknown", set()
    try:
        cur = {}
        with open("/proc/cpuinfo") as f:
            for line in f:
                if ":" in line:
                    k, v = [t.strip() for t in line.split(":", 1)]
                    cur[k] = v
                    if k == "model name":
                        model = v
                elif not line.strip():
                    if "physical id" in cur and "core id" in cur:
                        phys.add((cur["physical id"], cur["core id"]))
                    cur = {}
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, (len(phys) or None), usable


def _cpu_quota():
    """CPU limit of this container in cores (cgroup v2 cpu.max / v1 cfs quota), or None."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        if q != "max":
            return max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = int(f.read())
        if q > 0:
            return max(1, q // per)
    except (OSError, ValueError):
        pass
    return None


def _timed(fn, n_timed, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n_timed):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sum(ts) / len(ts), ts


def cpu_baseline(args):
    """The CPU oracle (a port of the reference's algorithm, oracle/unet_ref.py — the reference's files do not travel to the GPU
    box) on the host cores, SURVEY §8d / BASELINE.md §3: fp32, (a) the cfg-2 train step = DDPM.forward + backward + clip(1.0) +
    AdamW AT THE STATED BATCH (B = 64; --cpu-batch reduces it and scales linearly), (b) the cfg-2 CFG sample step (n = 64, w = 2),
    (c) cfg-1 (MNIST net, B = 64, T = 400) train and sample steps in full.  The thread count is the best of a small sweep (the
    usable cores, and 64 / 32 when the box has more: 128 SMT threads oversubscribe oneDNN on this workload); 1 warm-up + >= 3
    timed iterations per leg at that count (~30 s of CPU work for the train leg at 16 threads)."""
    from oracle import unet_ref as O
    torch.manual_seed(0)
    model, phys, usable = _cpu_info()
    nf, S, Bs = args.n_feat, args.size, (args.cpu_batch or args.batch)
    P = _rand_state(O.context_unet_spec(3, nf, 4, args.bottleneck_k))
    params = []
    for k, v in P.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
            params.append(v)
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-5)
    sched = O.ddpm_schedules(1e-4, 0.02, 1000)
    x, c, am = synthetic_batch(Bs, S, 4, "cpu")

    def train_step():
        ts = torch.randint(1, 1001, (Bs,))
        noise = torch.randn_like(x)
        keep = torch.bernoulli(torch.full((Bs,), 0.9))
        opt.zero_grad()
        loss = O.ddpm_loss(P, sched, 1000, x, c, am, ts, noise, keep, True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()

    quota = _cpu_quota()
    cap = min(usable, quota) if quota else usable
    cands = sorted({t for t in (cap, min(phys or cap, cap), 64, 32, 16) if t and t <= cap}, reverse=True)
    sweep, t_begin = {}, time.perf_counter()
    for i, t in enumerate(cands):                     # 1 warm-up + 1 timed iteration per candidate, inside a time budget
        if i and time.perf_counter() - t_begin > 60:
            break
        torch.set_num_threads(t)
        sweep[t] = round(_timed(train_step, 1)[0], 3)
        print(f"[bench] cpu baseline: {t} threads -> {sweep[t]:.2f} s per B={Bs} train step", file=sys.stderr, flush=True)
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    n_it = max(3, args.cpu_iters)
    t_train, all_train = _timed(train_step, n_it, warm=0)
    print(f"[bench] cpu baseline: train leg {t_train:.2f} s/iter at {best} threads", file=sys.stderr, flush=True)
    scale = args.batch / Bs

    # (b) cfg-2 sample step: one iteration of DDPM.sample's loop (denoiser on the CFG-doubled batch + update), eval mode
    xs = torch.randn(Bs, 3, S, S)
    zs = [torch.randn(Bs, 3, S, S)]

    def sample_step():
        with torch.no_grad():
            O.ddpm_sample(P, sched, 1000, 4, xs, zs, 2.0, steps=1)
    t_samp, _ = _timed(sample_step, n_it)
    print(f"[bench] cpu baseline: sample leg {t_samp:.2f} s/iter", file=sys.stderr, flush=True)

    # (c) cfg-1: MNIST_script.py net, 28x28, F=64, T=400, B=64 (train) / n=60 (sample: labels cycle over 10 classes)
    PM = _rand_state(O.mnist_unet_spec(1, 64, 10, 7))
    pm = []
    for k, v in PM.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
            pm.append(v)
    optm = torch.optim.Adam(pm, lr=1e-4)
    sched_m = O.ddpm_schedules(1e-4, 0.02, 400)
    xm, cm = torch.rand(64, 1, 28, 28), torch.randint(0, 10, (64,))

    def mnist_train():
        ts = torch.randint(1, 401, (64,))
        noise = torch.randn_like(xm)
        drop = torch.bernoulli(torch.full((64,), 0.1))
        optm.zero_grad()
        O.mnist_ddpm_loss(PM, sched_m, 400, xm, cm, ts, noise, drop, True).backward()
        optm.step()
    t_mt, _ = _timed(mnist_train, n_it)
    xsm, zsm = torch.randn(60, 1, 28, 28), [torch.randn(60, 1, 28, 28)]

    def mnist_sample():
        with torch.no_grad():
            O.ddpm_sample(PM, sched_m, 400, 10, xsm, zsm, 2.0, steps=1, net=O.mnist_context_unet)
    t_ms, _ = _timed(mnist_sample, n_it)

    return {"value": round(1.0 / (t_train * scale), 5), "unit": "denoiser-steps/s (train, B=%d)" % args.batch, "cores": best,
            "kind": "port", "cpu_model": model, "physical_cores": phys, "usable_threads": usable, "cgroup_cpu_quota": quota,
            "thread_sweep_s_per_iter": {str(k): v for k, v in sweep.items()},
            "sample": f"oracle train step (fwd+bwd+clip+AdamW) fp32 at B={Bs}" + (f" x{scale:g} scaled" if scale != 1 else " (the stated batch, no scaling)")
                      + f", 1 warm-up + {n_it} timed iters at {best} threads ({', '.join('%.2f' % v for v in all_train)} s)",
            "train_s_per_iter_at_sample_batch": round(t_train, 3),
            "sample_step": {"value": round(1.0 / (t_samp * scale), 5), "unit": "denoiser-steps/s (CFG sample, n=%d, w=2)" % args.batch,
                            "sample": f"one sample() iteration at n={Bs} (denoiser batch {2 * Bs})" + (f" x{scale:g} scaled" if scale != 1 else "")
                                      + f", {n_it} timed iters, {t_samp:.2f} s/iter"},
            "cfg1_mnist": {"train_steps_per_s": round(1.0 / t_mt, 3), "sample_steps_per_s": round(1.0 / t_ms, 3),
                           "sample": f"MNIST net 28x28 F=64 T=400: train B=64 (fwd+bwd+Adam) {t_mt * 1e3:.0f} ms, sample n=60 (CFG batch 120) "
                                     f"{t_ms * 1e3:.0f} ms; {n_it} timed iters each, full size (no scaling)"}}


def _sysfs_gpu(local):
    """Shader clock / power cap of the card as user space can read them (None where the box hides them)."""
    import glob
    out = {}
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
    try:
        with open(cards[local]) as f:
            lv = [ln.strip() for ln in f if ln.strip()]
        out["sclk_levels_mhz"] = [int(ln.split(":")[1].strip().lower().replace("mhz", "").replace("*", "")) for ln in lv]
        cur = [ln for ln in lv if ln.endswith("*")]
        if cur:
            out["sclk_now_mhz"] = int(cur[0].split(":")[1].strip().lower().replace("mhz", "").replace("*", ""))
        base = os.path.dirname(cards[local])
        for hw in glob.glob(os.path.join(base, "hwmon", "hwmon*")):
            for key, fn in (("power_cap_w", "power1_cap"), ("power_now_w", "power1_average"), ("power_now_w", "power1_input")):
                try:
                    with open(os.path.join(hw, fn)) as f:
                        out.setdefault(key, round(int(f.read()) / 1e6, 1))
                except (OSError, ValueError):
                    pass
    except (OSError, IndexError, ValueError):
        pass
    return out or None


def calibration(dev, iters=300):
    """What THIS box does on three fixed probes, measured in this process right before the timed region, so that lines from
    different boxes of the pool can be put on one scale (boxes differ by +-6 %, a round's work is 6 %):
    (1) the 64x64 128->128 3x3 halo convolution alone (B = 64, bf16, 77.3 GFLOP), `iters` back-to-back launches on RANDOM operands —
    the dominant kernel at its power-limited clock; (2) the same on ALL-ZERO operands — same cycles, no data toggling, the clock the
    power limit leaves (the pair is the DVFS signature); (3) one 1-GiB fp32 device copy (dm_cast; 2 GiB of HBM
    traffic); plus shader clock / power cap from sysfs when readable."""
    from diffusionmodel_amd import ops
    B, H, Ci, Co, k = 64, 64, 128, 128, 3
    flops = 2.0 * B * H * H * Co * k * k * Ci
    geom = dict(dtype=torch.bfloat16, B=B, Hi=H, Wi=H, C1=Ci, C2=0, Hq=H, Wq=H, sy=1, sx=1, T=9, KW=3, ty=1, tx=1, oy0=-1, ox0=-1, Ho=H, Wo=H, N=Co)
    y = torch.empty(B, H, H, Co, device=dev, dtype=torch.bfloat16)
    out = {"probe_conv": "conv3x3 halo kernel, 64x64, 128->128, B=64, bf16, %d back-to-back launches" % iters}
    g = torch.Generator(device=dev).manual_seed(5)
    for tag in ("random", "zeros"):
        if tag == "random":
            x = torch.randn(B, H, H, Ci, device=dev, generator=g).bfloat16()
            w = (torch.randn(Co, k, k, Ci, device=dev, generator=g) / (Ci * k * k) ** 0.5).bfloat16()
        else:
            x, w = torch.zeros(B, H, H, Ci, device=dev, dtype=torch.bfloat16), torch.zeros(Co, k, k, Ci, device=dev, dtype=torch.bfloat16)
        for _ in range(20):
            ops._conv_call(x, None, w.data_ptr(), k * k * Ci, y, **geom)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops._conv_call(x, None, w.data_ptr(), k * k * Ci, y, **geom)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        out["conv_%s_us" % tag] = round(us, 2)
        out["conv_%s_tflops" % tag] = round(flops / (us * 1e-6) / 1e12, 1)
        if tag == "random":
            # the same launch SUSTAINED: ~0.4 s more of it, then 1000 timed launches — the clock a box settles at under a long MFMA load
            # (boxes whose 300-launch figures agree to 1 % run the step's MFMA kernels 5 % apart: r04)
            for _ in range(5000):
                ops._conv_call(x, None, w.data_ptr(), k * k * Ci, y, **geom)
            e0.record()
            for _ in range(1000):
                ops._conv_call(x, None, w.data_ptr(), k * k * Ci, y, **geom)
            e1.record()
            torch.cuda.synchronize()
            out["conv_random_sustained_us"] = round(e0.elapsed_time(e1), 2)        # ms per 1000 launches = us per launch
    # (4) the same convolution ALTERNATING with the BatchNorm + GELU pass over its output (MFMA-bound / HBM-bound, as in the step): the
    # clock a box holds under the step's mixed load differs more between boxes than under either kernel alone
    x = torch.randn(B, H, H, Ci, device=dev, generator=g).bfloat16()
    w = (torch.randn(Co, k, k, Ci, device=dev, generator=g) / (Ci * k * k) ** 0.5).bfloat16()
    a = torch.empty_like(y)
    mean, rstd, gamma, beta = torch.zeros(Co, device=dev), torch.ones(Co, device=dev), torch.ones(Co, device=dev), torch.zeros(Co, device=dev)

    def pair():
        ops._conv_call(x, None, w.data_ptr(), k * k * Ci, y, **geom)
        ops.call("dm_bn_act_fwd", ops.ptr(y), ops.ptr(a), ops.L.DM_BF16, B * H * H, Co, ops.ptr(mean), ops.ptr(rstd), ops.ptr(gamma), ops.ptr(beta), ops.L.ACT_GELU)
    for _ in range(10):
        pair()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters // 2):
        pair()
    e1.record()
    torch.cuda.synchronize()
    out["mixed_conv_bn_us"] = round(e0.elapsed_time(e1) * 1e3 / (iters // 2), 2)
    del x, w, y, a
    n = 1 << 28
    a, b = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
    a.fill_(1.0)
    for _ in range(2):
        ops.call("dm_cast", ops.ptr(a), ops.ptr(b), ops.L.DM_F32, ops.L.DM_F32, n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.call("dm_cast", ops.ptr(a), ops.ptr(b), ops.L.DM_F32, ops.L.DM_F32, n)
    e1.record()
    torch.cuda.synchronize()
    out["copy_1GiB_TBps"] = round(5 * 8.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e12, 3)
    del a, b
    torch.cuda.empty_cache()
    out["sysfs"] = _sysfs_gpu(dev.index or 0)
    return out


def _calibration_ref():
    """The calibration block of the box the committed rocprof profile was taken on (profiles/r04_calibration_ref.json)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r04_calibration_ref.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


FAMILIES = (   # (family, substrings of kernel names) — first match wins
    ("conv", ("conv3x3_halo_kernel", "conv3x3_halo_pkernel", "conv3x3_packtap_kernel", "conv3x3_narrow_kernel", "conv_tap4_halo_kernel", "conv_pw_kernel",
              "conv_igemm", "splitk_epilogue")),
    ("wgrad", ("wgrad3x3_halo_kernel", "wgrad3x3_skinny_kernel", "wgrad_skinny_reduce", "wgrad_reduce", "wgrad_pw_kernel", "conv_wgrad")),
    ("batchnorm", ("bn_act_fwd", "bn_bwd_reduce", "bn_bwd_apply", "bn_finalize", "col_reduce", "col_stats")),
    ("optimiser", ("adamw", "sumsq", "pack_multi", "scatter_copy", "scaler_update", "pack_w")),
    ("attention", ("se_fwd", "se_bwd", "ca_z_", "ca_mix", "ca_bwd", "ca_gate", "ca_pix", "strip_reduce", "strip_fold", "scale_res", "dense_", "sgemm",
                   "act_fwd", "act_bwd", "sigmix", "colsum_small")),
    ("glue", ("upcat", "film", "gn_", "avgpool", "nchw", "nhwc", "qsample", "loss_", "randn", "draw_ts", "onehot", "cast_kernel", "add_kernel", "unpad",
              "maxpool", "cat_", "mask_axpy", "fill_t", "cfg_update")),
    ("torch", ("at::native", "_ZN2at", "rocclr")),
)


def family_of(name):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return "other"


def self_launch(argv, n):
    """`python bench.py --gpus N` outside torchrun: this parent makes NO GPU call (torch.cuda.device_count() does not initialise
    HIP on this image); it starts N fresh child processes — one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set —
    relays rank 0's JSON line and exits with the first non-zero child code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DM_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    import threading
    from diffusionmodel_amd.parallel import wait_ranks
    buf = []
    rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)     # drain rank 0's pipe while waiting
    rd.start()
    codes = wait_ranks(procs)                  # a rank that dies takes its siblings with it instead of leaving them in a collective
    rd.join(timeout=10)
    out = buf[0] if buf else ""
    line = [ln for ln in (out or "").splitlines() if ln.startswith("{")]
    if line:
        print(line[-1], flush=True)
    bad = [c for c in codes if c]
    if bad or not line:
        print(f"[bench] child exit codes {codes}", file=sys.stderr)
        raise SystemExit(bad[0] if bad else 1)


def launcher_selftest(args):
    """--launch-selftest: what a child of the self-launcher does with a gloo group on the CPU — joins, agrees on the world size and
    reports it.  Exists so that the launcher (ports, environment, relay of rank 0's line) is covered by a CPU test."""
    import torch.distributed as dist
    from diffusionmodel_amd import parallel
    rank, world, local = parallel.init_from_env("gloo", force=True)
    t = torch.ones(1)
    dist.all_reduce(t)
    devs = [None] * world
    dist.all_gather_object(devs, f"cpu:{local}")
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": args.gpus, "ranks": dist.get_world_size(), "sum": float(t), "devices": devs}), flush=True)
    dist.destroy_process_group()


def main():
    # stdout carries exactly ONE line, the JSON: libraries that print banners there (RCCL: "Librccl path : ...") go to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg5"],
                    help="cfg2 (default): BASELINE configs[1] — 64x64, n_feat 128, k 4, bf16, B 64 per GPU.  cfg5: BASELINE configs[4] — 128x128, "
                         "n_feat 256, k 8, fp16 (loss-scaled), B 8 per GPU.  The flags below override single values")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--n-feat", dest="n_feat", type=int, default=None)
    ap.add_argument("--bottleneck-k", dest="bottleneck_k", type=int, default=None)
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp32", "fp16"])
    ap.add_argument("--repeats", type=int, default=5, help="the timed region of --steps steps is run this many times; `value` is the median")
    ap.add_argument("--no-calibration", dest="calib", action="store_false")
    ap.add_argument("--no-dp-probe", dest="dp_probe", action="store_false", help="skip the in-process one-rank RCCL probe of the N=1 run")
    ap.add_argument("--sample-steps", dest="sample_steps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", dest="cpu", action="store_false")
    ap.add_argument("--cpu-batch", dest="cpu_batch", type=int, default=0, help="batch of the CPU-baseline legs (default: --batch, i.e. the stated B=64)")
    ap.add_argument("--cpu-iters", dest="cpu_iters", type=int, default=3)
    ap.add_argument("--launch-selftest", dest="launch_selftest", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one rank per GPU).  gloo is a REHEARSAL mode: the ranks may share devices (rank r uses device "
                         "r %% device_count; the library then selects its <= 64-KiB-LDS kernels), collectives go through the host — it "
                         "exercises the whole N-rank path of this script on a box with fewer GPUs, its numbers mean nothing")
    ap.add_argument("--buckets", type=int, default=6)
    ap.add_argument("--exec", dest="exec_mode", default="plan", choices=["plan", "graph", "eager"],
                    help="plan (default): the step is captured once and re-issued from C as plain launches (dm_plan_run; data parallel: "
                         "segments with the RCCL all-reduce between them); graph: replay the instantiated hipGraph (single process); "
                         "eager: every launch from Python")
    ap.add_argument("--graph", dest="graph", action="store_true", help="same as --exec graph")
    ap.add_argument("--shape-table", dest="shape_table", default="", help="write per-shape MFMA launch statistics to this file")
    ap.add_argument("--force-dp", dest="force_dp", action="store_true",
                    help="run the RCCL gradient all-reduce path even with one rank (rehearsal on a 1-GPU box)")
    args = ap.parse_args()
    preset = {"cfg2": dict(batch=64, size=64, n_feat=128, bottleneck_k=4, dtype="bf16"),
              "cfg5": dict(batch=8, size=128, n_feat=256, bottleneck_k=8, dtype="fp16")}[args.config]
    for k_, v_ in preset.items():
        if getattr(args, k_) is None:
            setattr(args, k_, v_)
    if args.config == "cfg5" and args.cpu and not args.cpu_batch:
        args.cpu_batch = 1                     # a B=8 cfg-5 train step is ~13 TFLOP on the CPU: one sample, scaled (stated in the line)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: be the launcher (no GPU call has been made in this process)
        if not args.launch_selftest:
            have = torch.cuda.device_count()
            if have < args.gpus and not (args.backend == "gloo" and have >= 1):
                raise SystemExit(f"--gpus {args.gpus} but only {have} HIP device(s) are visible")
        os.dup2(json_out.fileno(), 1)
        return self_launch(sys.argv[1:], args.gpus)
    if args.launch_selftest:
        os.dup2(json_out.fileno(), 1)
        return launcher_selftest(args)

    import diffusionmodel_amd as D
    from diffusionmodel_amd import ops, parallel
    if args.backend == "gloo":                 # rehearsal: ranks may share a device
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
    rank, world, local = parallel.init_from_env(args.backend, force=args.force_dp)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         "(or run `python bench.py --gpus N` without WORLD_SIZE set: it starts its own ranks)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    torch.manual_seed(0)                 # identical initial weights on every rank
    net = D.ContextUnet(3, args.n_feat, 4, bottleneck_k=args.bottleneck_k, dtype=dtype)
    ddpm = D.DDPM(net, (1e-4, 0.02), 1000, dev, drop_prob=0.1)
    ddpm.train()
    opt = D.FusedAdamW(ddpm.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, grad_scale=1.0 / world,
                       shadow_dtype=dtype if dtype != torch.float32 else torch.bfloat16)
    use_dp = world > 1 or args.force_dp
    # bucketed all-reduce of the flat gradient buffer, each bucket launched as soon as the backward pass has written it
    reducer = parallel.OverlappedGradReducer(opt, n_buckets=args.buckets) if use_dp else None
    if use_dp:
        parallel.broadcast_parameters(opt.flat_p)
        opt.refresh_shadow()                   # the masters changed outside step(): redo their bf16 shadow
        ops.bump_weight_epoch()
    # identical weights, but every rank draws its own timesteps / noise / keep-masks on its own data shard
    torch.manual_seed(1234 + rank)
    ddpm.rng_seed = 1234 + rank
    x, c, am = synthetic_batch(args.batch, args.size, 4, dev, seed=rank)

    # The step as a function of its static inputs; run eagerly it IS the eager step, captured it becomes the plan / graph.
    def body(st):
        opt.zero_grad()
        if reducer is not None:
            reducer.begin(capture=torch.cuda.is_current_stream_capturing())
        loss = ddpm(st.x, st.c, st.am)
        ddpm.scaler.scale(loss).backward()          # live in fp16 mode only (new_scripy.py:792)
        if reducer is not None:
            reducer.finish()
        ddpm.scaler.step(opt)                       # fp32 / bf16: plain opt.step()
        return loss

    mode = "graph" if args.graph else args.exec_mode
    if mode == "graph" and use_dp:
        raise SystemExit("--exec graph is single-process only (no collective inside a hipGraph here); use --exec plan")
    graphed = None
    if mode == "plan":
        graphed = D.GraphedTrainStep(ddpm, opt, x, c, am, mode="plan", body=body, runner=reducer.replay if reducer is not None else None)
    elif mode == "graph":
        graphed = D.GraphedTrainStep(ddpm, opt, x, c, am, mode="graph", body=body)

    class _Static:                              # the eager path's view of the inputs
        pass
    static = _Static()
    static.x, static.c, static.am = x, c, am

    def train_step(eager=False):
        if graphed is not None and not eager:
            return graphed()
        return body(static)

    def fence():
        if use_dp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        train_step(eager=(i == 0 and graphed is None))
    fence()
    # ---- calibration probes of THIS box, in this process, before the timed region (rank 0 reports; every rank runs them so the
    # ranks stay in step)
    calib = calibration(dev) if (args.calib and args.config == "cfg2") else None
    if calib is not None:
        for i in range(2):                     # back to the step's own cache / clock state
            train_step()
        fence()
    # Per-launch HIP events on the launch stream, recorded during the LAST timed step only (~540 event records per step cost ~2 ms
    # of wall time, which would otherwise distort `value`).  plan: dm_plan_run_timed puts the event pairs around the halo-kernel
    # launches of the replay; eager / graph: the last step is launched eagerly with torch events around every MFMA launch.
    planned = graphed is not None and graphed.plan is not None
    no_events = bool(os.environ.get("DM_BENCH_NO_EVENTS"))
    # The timed region — exactly --steps steps between fence() pairs — is run --repeats times; `value` is the MEDIAN region,
    # min / max go beside it.  The per-launch events are recorded in the last step of the LAST region only.
    regions, enq = [], []
    for rep in range(max(1, args.repeats)):
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            last = i == args.steps - 1 and rep == max(1, args.repeats) - 1 and not no_events
            if last and planned:
                plan = graphed.plan
                if reducer is not None:
                    graphed._runner = lambda pl: reducer.replay(pl, run=lambda a, b: pl.run_timed("halo_kernel|halo_pkernel", a, b))
                else:
                    graphed._runner = lambda pl: pl.run_timed("halo_kernel|halo_pkernel")
                loss = train_step()
                graphed._runner = reducer.replay if reducer is not None else None
                continue
            if last:
                ops.PROFILE_KINDS = ("conv_igemm", "conv_halo", "conv_wgrad") if dtype != torch.float32 else ("igemm_f32", "wgrad_f32")
                ops.PROFILE = []                   # (the instrumented step is launched eagerly: a replayed graph has no per-launch events)
            loss = train_step(eager=last)
        enq.append(time.perf_counter() - t0)   # host time to enqueue the steps (the GPU runs behind it; with a full queue this is back-pressure)
        fence()
        el = time.perf_counter() - t0
        if use_dp:                             # MAX over ranks, per region
            tt = torch.tensor([el], device=dev)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            el = float(tt.item())
        regions.append(el)
    elapsed = sorted(regions)[len(regions) // 2]
    t_enq = sorted(enq)[len(enq) // 2]
    prof, ops.PROFILE = ops.PROFILE or [], None
    prof_steps = 1                         # steps the event records cover
    timed = []                             # (kind, flops, seconds, shape)
    if planned and not no_events:
        res = graphed.plan.timed_results()
        meta_conv = [m for m in graphed.conv_meta if m[0] == "conv_halo"]
        meta_wg = [m for m in graphed.conv_meta if m[0] == "wgrad_halo"]
        got_conv = [r for r in res if "conv3x3_halo_kernel" in r[1] or "conv3x3_halo_pkernel" in r[1]]
        got_wg = [r for r in res if "wgrad3x3_halo_kernel" in r[1]]
        if len(got_conv) != len(meta_conv) or len(got_wg) != len(meta_wg):
            raise SystemExit(f"bench: the plan holds {len(got_conv)}/{len(got_wg)} halo conv/wgrad launches, the capture recorded {len(meta_conv)}/{len(meta_wg)}")
        timed = [(m[0], m[1], r[2] * 1e-3, m[2]) for m, r in zip(meta_conv, got_conv)] + [("conv_wgrad", m[1], r[2] * 1e-3, m[2]) for m, r in zip(meta_wg, got_wg)]
    else:
        timed = [(kind, flops, e0.elapsed_time(e1) * 1e-3, shape) for (kind, flops, e0, e1, shape) in prof]
    # host cost of ONE step on an idle queue (no back-pressure): what the launch path itself costs
    torch.cuda.synchronize()
    th0 = time.perf_counter()
    loss_extra = train_step()
    t_host_one = time.perf_counter() - th0
    torch.cuda.synchronize()
    if rank == 0:
        print(f"[bench] exec={mode}: host enqueue {t_enq / args.steps * 1e3:.2f} ms/step under load, {t_host_one * 1e3:.2f} ms for one step on an idle "
              f"queue; wall {elapsed / args.steps * 1e3:.2f} ms/step", file=sys.stderr)
    final_loss = float(loss.item())
    devices = [f"cuda:{local}"]
    if use_dp:
        devices = [None] * torch.distributed.get_world_size()
        torch.distributed.all_gather_object(devices, f"cuda:{local} ({torch.cuda.get_device_properties(local).name})")
    ranks_seen = torch.distributed.get_world_size() if use_dp else 1

    # ---- one MORE step, outside the timed region, with an event pair around EVERY kernel of the plan: per-family times, the
    # whole-step roofline and the HBM-bound table (the ~1100 event records of such a step cost milliseconds of wall time, so it
    # must not sit inside the timed region; the dominant kernel's events above do)
    fam_ms, fam_n, kern_ms, step_kernel_ms = {}, {}, {}, None
    if planned and not no_events:
        if reducer is not None:
            graphed._runner = lambda pl: reducer.replay(pl, run=lambda a, b: pl.run_timed("", a, b))
        else:
            graphed._runner = lambda pl: pl.run_timed("")
        train_step()
        graphed._runner = reducer.replay if reducer is not None else None
        allres = graphed.plan.timed_results(cap=16384)
        step_kernel_ms = 0.0
        for (_, nm, ms) in allres:
            fam = family_of(nm)
            fam_ms[fam] = fam_ms.get(fam, 0.0) + ms
            fam_n[fam] = fam_n.get(fam, 0) + 1
            short = nm.split("(")[0]
            for key in ("bn_bwd_apply", "bn_bwd_reduce", "bn_act_fwd", "adamw_kernel", "sumsq_kernel", "pack_multi", "wgrad_reduce"):
                if key in nm:
                    short = key
            k_ = kern_ms.setdefault(short, [0.0, 0])
            k_[0] += ms; k_[1] += 1
            step_kernel_ms += ms

    # ---- roofline of the dominant kernel family (implicit-GEMM conv fwd/dgrad + wgrad) from the events
    fl = {"conv_igemm": [0.0, 0.0, 0], "conv_wgrad": [0.0, 0.0, 0]}
    shapes = {}
    for (kind, flops, sec, shape) in timed:
        fl.setdefault(kind, [0.0, 0.0, 0])
        fl[kind][0] += flops
        fl[kind][1] += sec
        fl[kind][2] += 1
        sh = shapes.setdefault((kind, shape), [0.0, 0.0, 0])
        sh[0] += flops; sh[1] += sec; sh[2] += 1
    if args.shape_table and rank == 0:     # per-shape time / rate of the MFMA launches (tuning aid)
        with open(args.shape_table, "w") as f:
            for (kind, shape), (flo, sec, n) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{sec / prof_steps * 1e3:8.3f} ms/step {n / prof_steps:5.1f}x {sec / n * 1e6:8.1f} us {flo / sec / 1e12:8.1f} TF/s  {kind:11s} {shape}\n")
    if dtype == torch.float32:
        fl["conv_igemm"], fl["conv_wgrad"] = fl.get("igemm_f32", [0.0, 0.0, 0]), fl.get("wgrad_f32", [0.0, 0.0, 0])
    # dominant kernel: conv3x3_halo_kernel (forward + input gradient of the 3x3 layers) when the run used it, else the gather kernel
    fl.setdefault("conv_halo", [0.0, 0.0, 0])
    dom = "conv_halo" if fl["conv_halo"][2] else "conv_igemm"
    peak = PEAK_BF16_TFLOPS if dtype != torch.float32 else PEAK_F32_TFLOPS     # fp16 and bf16 MFMA run at the same dense rate
    ach = fl[dom][0] / max(fl[dom][1], 1e-12) / 1e12
    # algorithmic FLOPs / bytes of the captured step (ops.PROFILE_META: one record per MFMA launch and per BatchNorm pass)
    meta = graphed.conv_meta if graphed is not None else []
    flops_conv = sum(m[1] for m in meta if m[0] in ("conv_halo", "conv_igemm", "igemm_f32"))
    flops_wgrad = sum(m[1] for m in meta if m[0] in ("wgrad_halo", "conv_wgrad", "wgrad_f32"))
    flops_step = flops_conv + flops_wgrad
    ms_step = elapsed / args.steps * 1e3
    # algorithmic bytes of the dominant kernel's launches: input read once + weights + output written once
    esz = 4 if dtype == torch.float32 else 2
    alg_bytes = None
    if dom == "conv_halo" and meta:
        tot = 0.0
        for m in meta:
            if m[0] != "conv_halo":
                continue
            # shape string "B{B} {Hi}x{Wi} C{C1}+{C2} N{N} T{T} s{sy} t{ty}"
            tk = m[2].split()
            B_ = int(tk[0][1:]); Hi_, Wi_ = [int(v) for v in tk[1].split("x")]; C1_, C2_ = [int(v) for v in tk[2][1:].split("+")]
            N_ = int(tk[3][1:]); T_ = int(tk[4][1:])
            tot += esz * (B_ * Hi_ * Wi_ * (C1_ + C2_) + N_ * T_ * (C1_ + C2_) + B_ * Hi_ * Wi_ * N_)
        alg_bytes = tot / max(1, sum(1 for m in meta if m[0] == "conv_halo"))
    # HBM bytes per launch of the dominant kernel come from separate rocprofv3 --pmc passes over this same
    # command (they cannot be collected from inside the process); the committed summary is quoted when present
    traffic, traffic_src = None, None
    for cand in ("r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", cand)) as f:
                pmc = json.load(f)["kernels"].get("conv3x3_halo<bf16>" if dtype == torch.bfloat16 else "", None)
            if pmc and args.n_feat == 128 and args.size == 64 and args.batch == 64:
                traffic = pmc["hbm_bytes_per_launch"]
                traffic_src = f"profiles/{cand}: a QUOTE of separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command (scripts/pmc_collect.sh), not measured by this run"
                break
        except (OSError, KeyError, ValueError):
            continue

    def fam_rate(fam, flops):
        t = fam_ms.get(fam)
        return None if not t else round(flops / (t * 1e-3) / 1e12, 2)
    hbm_rows = []
    if fam_ms:
        n_par = opt.total
        alg = {"bn_bwd_apply": sum(m[1] for m in meta if m[0] == "bn_bwd_apply"), "bn_bwd_reduce": sum(m[1] for m in meta if m[0] == "bn_bwd_reduce"),
               "bn_act_fwd": sum(m[1] for m in meta if m[0] == "bn_fwd"), "adamw_kernel": 30.0 * n_par, "sumsq_kernel": 4.0 * n_par}
        cand = sorted(((k_, v_) for k_, v_ in kern_ms.items() if k_ in alg and alg[k_] > 0), key=lambda kv: -kv[1][0])[:4]
        for k_, (ms_, n_) in cand:
            hbm_rows.append({"kernel": k_, "launches": n_, "ms_per_step": round(ms_, 4), "algorithmic_bytes_per_step": int(alg[k_]),
                             "achieved_GBps": round(alg[k_] / (ms_ * 1e-3) / 1e9, 1), "frac_of_8TBps": round(alg[k_] / (ms_ * 1e-3) / 8e12, 4)})
    # the same fraction put on the scale of the box the committed rocprof profile was taken on: x (this box's probe time / that box's)
    ref_cal, frac_cal, frac_cal_note = _calibration_ref(), None, None
    if calib is not None and ref_cal and ref_cal.get("conv_random_us"):
        ratio = calib["conv_random_us"] / ref_cal["conv_random_us"]
        frac_cal = round(ach / peak * ratio, 4)
        frac_cal_note = ("frac x (this box's conv_random_us %.2f / the profile box's %.2f = %.3f): the dominant kernel's fraction as the box of "
                         "profiles/%s would have measured it" % (calib["conv_random_us"], ref_cal["conv_random_us"], ratio, ref_cal.get("profile", "r04_train_kernel_stats.csv")))
    roofline = {"bound": "mfma", "kernel": ("conv3x3_halo_kernel<%s> (dm_conv forward + input-gradient launches of the 3x3 layers)" if dom == "conv_halo"
                                            else "conv_igemm2_kernel<%s> (dm_conv forward + input-gradient launches)") % args.dtype,
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (avg)", "traffic_source": traffic_src,
                "traffic_kernel": "conv3x3_halo_kernel",
                "algorithmic_bytes_per_launch": None if alg_bytes is None else int(alg_bytes),
                "algorithmic_bytes_note": "average over this kernel's launches of: input read once + weights + output written once",
                "frac_calibrated": frac_cal, "frac_calibrated_note": frac_cal_note,
                "launches": fl[dom][2], "avg_launch_us": round(fl[dom][1] / max(fl[dom][2], 1) * 1e6, 2),
                "algorithmic_tflop_per_step": round(fl[dom][0] / prof_steps / 1e12, 4),
                "events": ("HIP event pairs on the launch stream around every halo-kernel launch of the LAST TIMED step (dm_plan_run_timed)") if planned else
                          "HIP events on the launch stream around every MFMA launch of the last timed step",
                "wgrad_halo": {"achieved": round(fl["conv_wgrad"][0] / max(fl["conv_wgrad"][1], 1e-12) / 1e12, 2), "launches": fl["conv_wgrad"][2],
                               "algorithmic_tflop_per_step": round(fl["conv_wgrad"][0] / prof_steps / 1e12, 4),
                               "note": "wgrad3x3_halo_kernel launches only (its reduce kernel is in families.wgrad)"},
                # the whole step against the same peak: every algorithmic FLOP of the step / the step time `value` is computed from
                "step": {"tflop": round(flops_step / 1e12, 4), "ms": round(ms_step, 3), "achieved": round(flops_step / (ms_step * 1e-3) / 1e12, 2),
                         "frac": round(flops_step / (ms_step * 1e-3) / 1e12 / peak, 4)},
                # one extra step AFTER the timed region with an event pair around every kernel of the plan
                "families": None if not fam_ms else {
                    "source": "one extra replay of the step outside the timed region, HIP event pair around every kernel (dm_plan_run_timed(\"\"))",
                    "kernel_ms_per_step": round(step_kernel_ms, 3), "launches": sum(fam_n.values()),
                    "ms": {k_: round(v_, 4) for k_, v_ in sorted(fam_ms.items(), key=lambda kv: -kv[1])}, "n": fam_n,
                    "all_dm_conv": {"achieved": fam_rate("conv", flops_conv), "algorithmic_tflop_per_step": round(flops_conv / 1e12, 4),
                                    "note": "every dm_conv kernel of the step: halo, four-tap, pointwise, gather, split-K epilogues"},
                    "all_wgrad": {"achieved": fam_rate("wgrad", flops_wgrad), "algorithmic_tflop_per_step": round(flops_wgrad / 1e12, 4),
                                  "note": "every weight-gradient kernel incl. the reduce launches"},
                    "non_mfma_ms": round(sum(v_ for k_, v_ in fam_ms.items() if k_ not in ("conv", "wgrad")), 4)},
                "hbm_bound": hbm_rows}

    # ---- CFG sampling rate (not part of `value`): every rank samples its shard of n = batch x world images (no exchange inside the
    # trajectory); steady-state step rate = difference of two runs, so the one-off graph capture of sample() cancels out
    sample = None
    if args.sample_steps > 0:
        ddpm.eval()
        n = args.batch

        def run(k):
            torch.cuda.synchronize()
            if use_dp:
                torch.distributed.barrier()
            t1 = time.perf_counter()
            parallel.sample_sharded(ddpm, n * world, (3, args.size, args.size), dev, guide_w=2.0, steps=k, seed=1, use_graph=True, gather=False)
            torch.cuda.synchronize()
            dt_ = torch.tensor([time.perf_counter() - t1], device=dev)
            if use_dp:
                torch.distributed.all_reduce(dt_, op=torch.distributed.ReduceOp.MAX)
            return float(dt_)
        try:                                  # the sampling rate is a side figure: a failure here must not lose the train metric
            run(2)                                                                               # warm-up / caches
            k0 = 4
            t_short = min(run(k0), run(k0))                # (two of each, the faster one: the capture cost varies from call to call)
            t_long = min(run(k0 + args.sample_steps), run(k0 + args.sample_steps))
            if t_long - t_short < 1e-4 * args.sample_steps:
                raise RuntimeError(f"sampling timer: {args.sample_steps} extra steps took {t_long - t_short:.4f} s more than {k0} steps "
                                   f"({t_short:.3f} s vs {t_long:.3f} s): capture cost dominates, raise --sample-steps")
            rate = args.sample_steps / (t_long - t_short)
            sample = {"steps_per_s": round(rate, 3), "n": n * world, "denoiser_batch_per_gpu": 2 * n, "guide_w": 2.0, "hipgraph": True,
                      "encoder_dedup": True, "steps_timed": args.sample_steps, "images_x_steps_per_s": round(rate * n * world, 1)}
            if flops_step:
                # forward FLOPs per sample = a third of the train step's; with the encoder + up0 (51.6 % of the MACs) computed
                # once for the two CFG halves a step is n * (0.516 + 2 * 0.484) forwards
                fwd = flops_step / 3.0 / args.batch
                tfl = fwd * n * (0.516 + 2 * 0.484) / 1e12
                sample["algorithmic_tflop_per_step_per_gpu"] = round(tfl, 4)
                sample["achieved_tflops_per_gpu"] = round(tfl * rate, 2)
                sample["frac_of_mfma_peak"] = round(tfl * rate / peak, 4)
        except Exception as exc:              # noqa: BLE001 — reported in the JSON
            sample = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        ddpm.train()

    # ---- data-parallel facts of this run: what the N > 1 curve is read against
    dp = None
    if use_dp and planned:
        reducer.time_tail = True
        tails = []
        for _ in range(5):
            train_step()
            tails.append(reducer.tail_wait_ms())
        reducer.time_tail = False
        dp = {"ranks": ranks_seen, "buckets": len(reducer.buckets), "allreduce_bytes_per_step_per_rank": reducer.bytes_per_step(),
              "bucket_bytes": [4 * (b["hi"] - b["lo"]) for b in reducer.buckets],
              "tail_wait_ms": round(sorted(tails)[len(tails) // 2], 3),
              "tail_wait_note": "launch-stream time between the last backward kernel and all bucket all-reduces done (event pair in "
                                "OverlappedGradReducer.replay), median of 5 steps after the timed regions: the part of the reduction the backward pass did not hide",
              "collectives_on": "side stream" if reducer._stream is not None else "launch stream -> RCCL's own stream", "force_dp": bool(args.force_dp)}
    elif world == 1 and planned and args.dp_probe and args.config == "cfg2":
        # plain N = 1 (the line's value) next to the SAME process driving the one-rank RCCL path (--force-dp): the floor under
        # every point of the N > 1 curve.  Built after the timed regions; a failure here must not lose the line.
        try:
            parallel.init_from_env("nccl", force=True)
            red2 = parallel.OverlappedGradReducer(opt, n_buckets=args.buckets)

            def body2(st):
                opt.zero_grad()
                red2.begin(capture=torch.cuda.is_current_stream_capturing())
                loss_ = ddpm(st.x, st.c, st.am)
                ddpm.scaler.scale(loss_).backward()
                red2.finish()
                ddpm.scaler.step(opt)
                return loss_
            g2 = D.GraphedTrainStep(ddpm, opt, x, c, am, mode="plan", body=body2, runner=red2.replay)

            def region(fn):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t1) / args.steps * 1e3
            for _ in range(3):
                g2()
            a_, b_ = [], []
            for _ in range(3):                 # alternate, same box, same minute
                a_.append(region(graphed))
                b_.append(region(g2))
            red2.time_tail = True
            g2()
            dp = {"ranks": 1, "n1_plain_ms_per_step": [round(v, 3) for v in a_], "n1_force_dp_ms_per_step": [round(v, 3) for v in b_],
                  "force_dp_overhead_ms": round(sorted(b_)[1] - sorted(a_)[1], 3), "buckets": len(red2.buckets),
                  "allreduce_bytes_per_step_per_rank": red2.bytes_per_step(), "tail_wait_ms": round(red2.tail_wait_ms(), 3),
                  "note": "in-process probe after the timed regions: the planned step re-captured with OverlappedGradReducer over a one-rank RCCL "
                          "group, alternated three times with the plain plan (%d steps per region)" % args.steps}
            red2.time_tail = False
            torch.distributed.destroy_process_group()
        except Exception as exc:              # noqa: BLE001 — reported in the JSON
            dp = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    cpu = None
    if rank == 0 and world == 1 and args.cpu:
        cpu = cpu_baseline(args)

    if rank == 0:
        value = world * args.steps / elapsed
        out = {"metric": f"denoiser-steps/sec (train fwd+bwd+clip+AdamW, {args.size}x{args.size}, B={args.batch}/GPU)", "value": round(value, 4),
               "unit": "steps/s", "n_gpus": world, "ranks": ranks_seen, "devices": devices, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 3),
               "repeats": {"n": len(regions), "of_steps": args.steps, "value_is": "median region",
                           "ms_per_step_all": [round(r / args.steps * 1e3, 3) for r in regions],
                           "ms_per_step_min": round(min(regions) / args.steps * 1e3, 3), "ms_per_step_max": round(max(regions) / args.steps * 1e3, 3)},
               "calibration": calib, "calibration_ref": ref_cal,
               "ms_per_step_calibrated": (None if not (calib and ref_cal and ref_cal.get("conv_random_us")) else
                                          round(elapsed / args.steps * 1e3 * ref_cal["conv_random_us"] / calib["conv_random_us"], 3)),
               "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "new_scripy.py ContextUnet %dx%d n_feat=%d T=1000 %s, train step batch=%d per GPU (BASELINE configs[%d])"
                                      % (args.size, args.size, args.n_feat, args.dtype, args.batch, 4 if args.config == "cfg5" else (1 if world == 1 else 2)),
                          "preset": args.config,
                          "global_batch": args.batch * world, "bottleneck_k": args.bottleneck_k, "n_classes": 4,
                          "parallelism": "dp%d" % world, "backend": args.backend if use_dp else None, "exec": mode, "host_ms_one_step_idle_queue": round(t_host_one * 1e3, 3), "samples_per_s": round(value * args.batch, 2)},
               "loss": final_loss, "roofline": roofline, "cpu_baseline": cpu, "sample": sample, "dp": dp}
        print(json.dumps(out), file=json_out, flush=True)
    if use_dp:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
