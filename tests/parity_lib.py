"""Measurements behind the bf16 / training-trajectory parity tests (tests/test_gpu_bf16.py) and behind
profiles/r02_parity.json (scripts/parity_report.py writes what these functions return).  TEST INFRASTRUCTURE.

Yardsticks:
* tests/golden/bf16_autocast.npz — the REFERENCE run under torch.autocast("cpu", bfloat16) next to the same code in float64
  (make_golden.py:gen_bf16_autocast): its eps MSE, loss and per-child gradient norm / cosine say how far bf16 moves the
  reference itself; the HIP bf16 path is held to a small multiple of that;
* tests/golden/train3.npz — three optimiser steps of the reference's train loop (accumulation 2, clip 1.0, torch AdamW) on
  injected draws;
* the CPU oracle (oracle/unet_ref.py, pinned to the reference by the other fixtures) in float64 / float32 for per-parameter
  gradients and for the F=128 case no fixture covers.
"""
import json
import os

import numpy as np
import torch

from oracle import synth, unet_ref as O

DEV = "cuda:0"
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCHEMA = json.load(open(os.path.join(G, "schema.json")))
si = synth.synth_input


def npz(name):
    return np.load(os.path.join(G, name + ".npz"))


def _cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def child_grad_vectors(net, prefix=""):
    """{top-level child: flat float64 gradient} of an nn.Module after backward (missing gradients count as zeros)."""
    out = {}
    for cn, ch in net.named_children():
        vs = [(p.grad if p.grad is not None else torch.zeros_like(p)).detach().double().reshape(-1).cpu() for p in ch.parameters()]
        if vs:
            out[cn] = torch.cat(vs).numpy()
    return out


def oracle_child_vectors(P, names_by_child):
    return {cn: np.concatenate([(P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])).double().reshape(-1).numpy() for k in ks])
            for cn, ks in names_by_child.items()}


def _names_by_child(net, prefix=""):
    d = {}
    for cn, ch in net.named_children():
        ks = [f"{prefix}{cn}.{k}" for k, _ in ch.named_parameters()]
        if ks:
            d[cn] = ks
    return d


def synth_state(tag, dtype=torch.float32, grad=True):
    P = {}
    for k, s in SCHEMA[tag]:
        v = synth.synth_tensor(k, tuple(s))
        if v.is_floating_point():
            v = v.to(dtype)
            if grad and "running" not in k:
                v.requires_grad_(True)
        P[k] = v
    return P


def hip_unet(tag, dtype, k=4, F=32):
    import diffusionmodel_amd as D
    net = D.ContextUnet(3, F, 4, bottleneck_k=k, dtype=dtype)
    net.load_state_dict({kk: synth.synth_tensor(kk, tuple(s)) for kk, s in SCHEMA[tag]}, strict=True)
    return net.to(DEV)


def grad_table(hip_vecs, ref_vecs, skip=("local_enhance",)):
    """Per child: norm ratio error |hip|/|ref| - 1 and 1 - cosine."""
    t = {}
    for cn, r in ref_vecs.items():
        if cn in skip or np.linalg.norm(r) == 0:
            continue
        h = hip_vecs[cn]
        t[cn] = {"norm_rel_err": float(np.linalg.norm(h) / np.linalg.norm(r) - 1.0), "one_minus_cos": float(1.0 - _cos(h, r))}
    return t


def _autocast_fixture(dtype):
    """The reference's autocast run that is the yardstick for `dtype` (float32 runs are listed next to the bf16 one)."""
    return npz("fp16_autocast" if dtype == torch.float16 else "bf16_autocast")


def _loss_scale(dtype):
    """float16 gradients are produced under the GradScaler's initial scale and unscaled afterwards (new_scripy.py:792-797)."""
    return 65536.0 if dtype == torch.float16 else 1.0


def unet_case(dtype, modes=("eval", "train")):
    """HIP ContextUnet(F=32) at 64x64 on the unet32_64 inputs against the reference's float64 run."""
    tag = "unet32_64"
    g, gb = npz(tag), _autocast_fixture(dtype)
    ls = _loss_scale(dtype)
    x = si(tag + ".x", (2, 3, 64, 64))
    c, t, mk = torch.tensor(g["c"]), torch.tensor(g["t"]), torch.tensor(g["ctx_mask"])
    probe = si(tag + ".probe", (2, 3, 64, 64))
    out = {}
    for mode in modes:
        train = mode == "train"
        net = hip_unet(tag, dtype)
        net.train(train)
        eps = net(x.to(DEV), c.to(DEV), t.to(DEV), mk.to(DEV))
        loss = (eps * probe.to(DEV)).mean()
        (loss * ls).backward()
        e = eps.detach().cpu().double().numpy()
        e64 = gb[f"unet.{mode}.eps64"]
        # float64 oracle gradients (the oracle equals the reference in float64 to 3e-13, tests/test_oracle_golden.py)
        P = synth_state(tag, torch.float64)
        eo = O.context_unet(P, x.double(), c, t.double(), mk.double(), train)
        (eo * probe.double()).mean().backward()
        ref_vecs = oracle_child_vectors(P, _names_by_child(net))
        out[mode] = {
            "eps_mse_vs_ref64": float(((e - e64) ** 2).mean()), "eps_maxabs_vs_ref64": float(np.abs(e - e64).max()),
            "signal_power": float((e64 ** 2).mean()),
            "ref_autocast_bf16_mse": float(gb[f"unet.{mode}.mse_bf16_vs_64"]), "ref_autocast_bf16_maxabs": float(gb[f"unet.{mode}.maxabs_bf16_vs_64"]),
            "loss": float(loss.item()), "loss_ref64": float(gb[f"unet.{mode}.loss64"]), "loss_ref_autocast_bf16": float(gb[f"unet.{mode}.loss_bf16"]),
            "probe_power": float((probe ** 2).mean()), "n_elements": int(probe.numel()),
            "grads": grad_table({cn: v / ls for cn, v in child_grad_vectors(net).items()}, ref_vecs),
            "ref_autocast_bf16_grads": {cn: {"norm_rel_err": float(gb[f"unet.{mode}.gn_bf16.{cn}"] / gb[f"unet.{mode}.gn64.{cn}"] - 1.0),
                                             "one_minus_cos": float(1.0 - gb[f"unet.{mode}.cos_bf16.{cn}"])}
                                        for cn in ref_vecs if cn != "local_enhance" and float(gb[f"unet.{mode}.gn64.{cn}"]) > 0},
        }
    return out


def ddpm_case(dtype):
    """HIP DDPM.forward on the ddpm_fwd64 case (draws injected) against the reference's float64 run."""
    import diffusionmodel_amd as D
    tag, B, S, n_T = "ddpm_fwd64", 4, 64, 1000
    g, gb = npz(tag), _autocast_fixture(dtype)
    ls = _loss_scale(dtype)
    x = si(tag + ".x", (B, 3, S, S))
    c = torch.tensor([(i + 2) % 4 for i in range(B)])
    am = synth.synth_attn_mask(B, S)
    ts, keep = torch.tensor(g["ts"]), torch.tensor(g["keep"])
    noise = synth.synth_noise(tag + ".noise", (B, 3, S, S))
    out = {}
    for mode in ("train", "eval"):
        ddpm = D.DDPM(D.ContextUnet(3, 32, 4, bottleneck_k=4, dtype=dtype), (1e-4, 0.02), n_T, DEV, drop_prob=0.1)
        sd = {k: synth.synth_tensor(k, tuple(s)) for k, s in SCHEMA[tag]}
        for k in D.SCHEDULE_KEYS:
            sd[k] = D.ddpm_schedules(1e-4, 0.02, n_T)[k]
        ddpm.load_state_dict(sd)
        ddpm.train(mode == "train")
        loss = ddpm(x.to(DEV), c.to(DEV), am.to(DEV), ts=ts.to(DEV), noise=noise.to(DEV), ctx_mask=keep.to(DEV))
        rec = {"loss": float(loss.item()), "loss_ref64": float(gb[f"ddpm.{mode}.loss64"]), "loss_ref_autocast_bf16": float(gb[f"ddpm.{mode}.loss_bf16"])}
        if mode == "train":
            ddpm.scaler.scale(loss).backward()         # float16: x 2^16 (DmGradScaler's initial scale); fp32 / bf16: pass-through
            assert ddpm.scaler.get_scale() == ls
            P = synth_state(tag, torch.float64)
            sched = {k: v.double() for k, v in O.ddpm_schedules(1e-4, 0.02, n_T).items()}
            lo = O.ddpm_loss(P, sched, n_T, x.double(), c, am.double(), ts, noise.double(), keep.double(), True)
            lo.backward()
            rec["loss_oracle64"] = float(lo.item())
            ref_vecs = oracle_child_vectors(P, _names_by_child(ddpm.nn_model, "nn_model."))
            rec["grads"] = grad_table({cn: v / ls for cn, v in child_grad_vectors(ddpm.nn_model).items()}, ref_vecs)
            rec["ref_autocast_bf16_grads"] = {cn: {"norm_rel_err": float(gb[f"ddpm.train.gn_bf16.{cn}"] / gb[f"ddpm.train.gn64.{cn}"] - 1.0),
                                                   "one_minus_cos": float(1.0 - gb[f"ddpm.train.cos_bf16.{cn}"])}
                                              for cn in ref_vecs if cn != "local_enhance" and float(gb[f"ddpm.train.gn64.{cn}"]) > 0}
        out[mode] = rec
    return out


def f128_case(B=8, S=64, F=128):
    """The benchmark's width (F=128, 64x64, k=4) at B=8 in bf16 against the float32 oracle on the same seeded weights and
    inputs; the reference's autocast run does not exist for this size (no fixture), so the yardstick is the oracle itself run
    under torch.autocast("cpu", bfloat16) — same ops as the reference under autocast."""
    import diffusionmodel_amd as D
    torch.manual_seed(1234)
    net = D.ContextUnet(3, F, 4, bottleneck_k=4, dtype=torch.bfloat16)
    with torch.no_grad():                                     # non-trivial BatchNorm state so that eval mode is not the identity
        for n_, b in net.named_buffers():
            if n_.endswith("running_mean"):
                b.normal_(0, 0.1)
            elif n_.endswith("running_var"):
                b.uniform_(0.6, 1.4)
    sd = {k: v.detach().clone().cpu() for k, v in net.state_dict().items()}
    net = net.to(DEV)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 3, S, S, generator=g).clamp_(-1, 1)
    c = torch.randint(0, 4, (B,), generator=g)
    t = torch.rand(B, generator=g)
    mk = (torch.rand(B, generator=g) > 0.2).float()
    probe = torch.randn(B, 3, S, S, generator=g)
    out = {}
    for mode in ("eval", "train"):
        train = mode == "train"
        net.load_state_dict(sd)
        net.train(train)
        eps = net(x.to(DEV), c.to(DEV), t.to(DEV), mk.to(DEV))
        loss = (eps * probe.to(DEV)).mean()
        loss.backward()
        hip_vecs = child_grad_vectors(net)
        net.zero_grad()
        e = eps.detach().cpu().double().numpy()
        P = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and "running" not in k) else v.clone()) for k, v in sd.items()}
        eo = O.context_unet(P, x, c, t, mk, train)
        lo = (eo * probe).mean()
        lo.backward()
        ref_vecs = oracle_child_vectors(P, _names_by_child(net))
        e32 = eo.detach().double().numpy()
        P16 = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and "running" not in k) else v.clone()) for k, v in sd.items()}
        with torch.autocast("cpu", dtype=torch.bfloat16):
            eo16 = O.context_unet(P16, x, c, t, mk, train)
            lo16 = (eo16.float() * probe).mean()
        lo16.backward()
        e16 = eo16.detach().float().double().numpy()
        a16_vecs = oracle_child_vectors(P16, _names_by_child(net))
        out[mode] = {"eps_mse_vs_oracle32": float(((e - e32) ** 2).mean()), "eps_maxabs_vs_oracle32": float(np.abs(e - e32).max()),
                     "signal_power": float((e32 ** 2).mean()),
                     "oracle_autocast_bf16_mse": float(((e16 - e32) ** 2).mean()), "oracle_autocast_bf16_maxabs": float(np.abs(e16 - e32).max()),
                     "loss": float(loss.item()), "loss_oracle32": float(lo.item()), "loss_oracle_autocast_bf16": float(lo16.item()),
                     "probe_power": float((probe ** 2).mean()), "n_elements": int(probe.numel()),
                     "grads": grad_table(hip_vecs, ref_vecs), "oracle_autocast_bf16_grads": grad_table(a16_vecs, ref_vecs)}
    return out


def f128_b2_case(dtype, modes=("eval", "train")):
    """The benchmark's width against the REFERENCE ITSELF: tests/golden/f128_b2.npz (make_golden.py:gen_f128_b2 — the imported
    reference at n_feat = 128, 64x64, k = 4, B = 2 in float64 / float32 / autocast(bfloat16)).  eps against the float64 run, the
    probe loss, per-child gradient NORMS (the 106 M-element gradient vectors are not stored) against the float64 norms."""
    import diffusionmodel_amd as D
    tag, B, S, F = "f128_b2", 2, 64, 128
    g = npz(tag)
    ls = _loss_scale(dtype)
    spec = O.context_unet_spec(3, F, 4, 4)
    sd = synth.synth_state(spec)
    x = si(tag + ".x", (B, 3, S, S))
    c, t, mk = torch.tensor(g["c"]), torch.tensor(g["t"]), torch.tensor(g["ctx_mask"])
    probe = si(tag + ".probe", (B, 3, S, S))
    out = {}
    for mode in modes:
        net = D.ContextUnet(3, F, 4, bottleneck_k=4, dtype=dtype)
        net.load_state_dict(sd, strict=True)
        net = net.to(DEV)
        net.train(mode == "train")
        eps = net(x.to(DEV), c.to(DEV), t.to(DEV), mk.to(DEV))
        loss = (eps * probe.to(DEV)).mean()
        (loss * ls).backward()
        e = eps.detach().cpu().double().numpy()
        e64 = g[f"{mode}.eps64"]
        norms = {cn: float(np.linalg.norm(v)) / ls for cn, v in child_grad_vectors(net).items()}
        out[mode] = {
            "eps_mse_vs_ref64": float(((e - e64) ** 2).mean()), "eps_maxabs_vs_ref64": float(np.abs(e - e64).max()),
            "signal_power": float(g[f"{mode}.power64"]), "ref_fp32_maxabs": float(g[f"{mode}.eps32_maxabs_vs_64"]),
            "ref_autocast_bf16_mse": float(g[f"{mode}.mse_bf16_vs_64"]), "ref_autocast_bf16_maxabs": float(g[f"{mode}.maxabs_bf16_vs_64"]),
            "loss": float(loss.item()), "loss_ref64": float(g[f"{mode}.loss64"]), "loss_ref_autocast_bf16": float(g[f"{mode}.loss_bf16"]),
            "probe_power": float((probe ** 2).mean()), "n_elements": int(probe.numel()),
            "grad_norm_rel_err": {cn: norms[cn] / float(g[f"{mode}.gn64.{cn}"]) - 1.0 for cn in norms
                                  if cn != "local_enhance" and float(g[f"{mode}.gn64.{cn}"]) > 0},
            "ref_autocast_grad_norm_rel_err": {cn: float(g[f"{mode}.gn_bf16.{cn}"] / g[f"{mode}.gn64.{cn}"] - 1.0) for cn in norms
                                               if cn != "local_enhance" and float(g[f"{mode}.gn64.{cn}"]) > 0},
            "ref_autocast_one_minus_cos": {cn: float(1.0 - g[f"{mode}.cos_bf16.{cn}"]) for cn in norms if f"{mode}.cos_bf16.{cn}" in g.files},
        }
        del net
    return out


def f128_b2_band_case(dtype=torch.bfloat16, modes=("eval", "train")):
    """Per-child gradient-norm errors of the HIP path and of the REFERENCE's autocast(bfloat16) run as DISTRIBUTIONS: the f128_b2
    case on its own input (j = 0, f128_b2.npz) and on K = 5 inputs 1e-3 of noise away from it (f128_b2_band.npz,
    make_golden.py:gen_f128_b2_band — reference float64 and autocast norms per input).  One input is one rounding-noise realisation:
    the reference's own worst-child error over these six inputs spans 0.023 .. 0.138 in eval mode."""
    import diffusionmodel_amd as D
    tag, B, S, F = "f128_b2", 2, 64, 128
    g0, gb = npz(tag), npz(tag + "_band")
    K = int(gb["K"])
    ls = _loss_scale(dtype)
    spec = O.context_unet_spec(3, F, 4, 4)
    sd = synth.synth_state(spec)
    x0 = si(tag + ".x", (B, 3, S, S))
    c, t, mk = torch.tensor(g0["c"]), torch.tensor(g0["t"]), torch.tensor(g0["ctx_mask"])
    probe = si(tag + ".probe", (B, 3, S, S))
    out = {}
    for mode in modes:
        hip_err, ref_err = [], []
        net = D.ContextUnet(3, F, 4, bottleneck_k=4, dtype=dtype)
        for j in range(K + 1):
            x = x0 if j == 0 else x0 + float(gb["pert_scale"]) * synth.synth_noise(f"{tag}.pert{j}", (B, 3, S, S))
            key = (lambda q, cn: f"{mode}.{q}.{cn}") if j == 0 else (lambda q, cn, j=j: f"{j}.{mode}.{q}.{cn}")
            g = g0 if j == 0 else gb
            net.load_state_dict(sd, strict=True)
            net = net.to(DEV)
            net.train(mode == "train")
            net.zero_grad()
            eps = net(x.to(DEV), c.to(DEV), t.to(DEV), mk.to(DEV))
            ((eps * probe.to(DEV)).mean() * ls).backward()
            norms = {cn: float(np.linalg.norm(v)) / ls for cn, v in child_grad_vectors(net).items()}
            kids = [cn for cn in norms if cn != "local_enhance" and key("gn64", cn) in g.files and float(g[key("gn64", cn)]) > 0]
            hip_err.append({cn: norms[cn] / float(g[key("gn64", cn)]) - 1.0 for cn in kids})
            ref_err.append({cn: float(g[key("gn_bf16", cn)]) / float(g[key("gn64", cn)]) - 1.0 for cn in kids})
        kids = list(hip_err[0])
        rms = lambda rows: float(np.sqrt(np.mean([v * v for r in rows for v in r.values()])))
        out[mode] = {"inputs": K + 1, "hip_worst_child_per_input": [max(abs(v) for v in r.values()) for r in hip_err],
                     "ref_autocast_worst_child_per_input": [max(abs(v) for v in r.values()) for r in ref_err],
                     "hip_pooled_rms": rms(hip_err), "ref_autocast_pooled_rms": rms(ref_err),
                     "hip_rms_per_child": {cn: float(np.sqrt(np.mean([r[cn] ** 2 for r in hip_err]))) for cn in kids},
                     "ref_autocast_rms_per_child": {cn: float(np.sqrt(np.mean([r[cn] ** 2 for r in ref_err]))) for cn in kids}}
        del net
    return out


def cfg2_full_size_case(B=64, S=64, F=128):
    """BASELINE configs[1] at its stated size (64x64, n_feat = 128, k = 4, B = 64, bf16), train mode, forward only: eps of the HIP
    path and the DDPM.forward loss (draws injected) against the float32 oracle on the same weights; yardstick = the oracle under
    torch.autocast("cpu", bfloat16).  Forward only so that the CPU side stays at a few seconds on the box's 16 threads."""
    import diffusionmodel_amd as D
    torch.manual_seed(4321)
    net = D.ContextUnet(3, F, 4, bottleneck_k=4, dtype=torch.bfloat16)
    sd = {k: v.detach().clone().cpu() for k, v in net.state_dict().items()}
    ddpm = D.DDPM(net, (1e-4, 0.02), 1000, DEV, drop_prob=0.1)
    ddpm.train()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 3, S, S, generator=g).clamp_(-1, 1)
    c = torch.randint(0, 4, (B,), generator=g)
    ts = torch.randint(1, 1001, (B,), generator=g)
    keep = (torch.rand(B, generator=g) > 0.1).float()
    noise = torch.randn(B, 3, S, S, generator=g)
    am = synth.synth_attn_mask(B, S)
    t = ts.float() / 1000
    with torch.no_grad():
        eps = ddpm.nn_model(x.to(DEV), c.to(DEV), t.to(DEV), keep.to(DEV)).cpu().double().numpy()
        ddpm.load_state_dict({**{"nn_model." + k: v for k, v in sd.items()}, **{k: getattr(ddpm, k) for k in D.SCHEDULE_KEYS}})
        loss = float(ddpm(x.to(DEV), c.to(DEV), am.to(DEV), ts=ts.to(DEV), noise=noise.to(DEV), ctx_mask=keep.to(DEV)))
    P = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        e32 = O.context_unet(P, x, c, t, keep, True).double().numpy()
        P = {k: v.clone() for k, v in sd.items()}
        with torch.autocast("cpu", dtype=torch.bfloat16):
            e16 = O.context_unet(P, x, c, t, keep, True).float().double().numpy()
        PD = {"nn_model." + k: v.clone() for k, v in sd.items()}
        l32 = float(O.ddpm_loss(PD, O.ddpm_schedules(1e-4, 0.02, 1000), 1000, x, c, am, ts, noise, keep, True))
    return {"eps_mse_vs_oracle32": float(((eps - e32) ** 2).mean()), "oracle_autocast_bf16_mse": float(((e16 - e32) ** 2).mean()),
            "eps_maxabs_vs_oracle32": float(np.abs(eps - e32).max()), "oracle_autocast_bf16_maxabs": float(np.abs(e16 - e32).max()),
            "signal_power": float((e32 ** 2).mean()), "loss": loss, "loss_oracle32": l32, "B": B}


def sample_f128_case(n=16, steps=3, w=2.0, S=64, F=128, n_T=1000, seed=1234):
    """BASELINE configs[3] on the benchmarked kernel set (bf16, n_feat = 128, hipGraph replay, encoder de-dup, broadcast skip
    tensors, BatchNorm folded into the conv epilogues): `steps` CFG sampling steps (new_scripy.py:441-477) against the float32
    oracle on the same weights, start noise and per-step noise; yardstick = the oracle under torch.autocast("cpu", bfloat16).
    The graph draws its noise in-kernel (Philox: key = seed, counter high word = the step index i, low word = the element quad);
    the same numbers are regenerated with dm_randn_slice and injected into the oracle — and into an eager HIP run with `zs=`,
    which must agree with the replayed graph."""
    import diffusionmodel_amd as D
    from diffusionmodel_amd import ops
    torch.manual_seed(977)
    net = D.ContextUnet(3, F, 4, bottleneck_k=4, dtype=torch.bfloat16)
    with torch.no_grad():
        for n_, b in net.named_buffers():
            if n_.endswith("running_mean"):
                b.normal_(0, 0.1)
            elif n_.endswith("running_var"):
                b.uniform_(0.6, 1.4)
    sd = {k: v.detach().clone().cpu() for k, v in net.state_dict().items()}
    ddpm = D.DDPM(net, (1e-4, 0.02), n_T, DEV, drop_prob=0.1)
    ddpm.eval()
    shape = (n, 3, S, S)
    x_T = ops.randn(shape, DEV, seed, 0)
    zs = [ops.randn(shape, DEV, seed, n_T - j) for j in range(steps)]
    xg = ddpm.sample(n, (3, S, S), DEV, guide_w=w, use_graph=True, seed=seed, steps=steps).cpu().double().numpy()
    xe = ddpm.sample(n, (3, S, S), DEV, guide_w=w, x_T=x_T, zs=zs, steps=steps).cpu().double().numpy()
    xT, zc = x_T.cpu(), [z.cpu() for z in zs]
    sched = O.ddpm_schedules(1e-4, 0.02, n_T)
    PD = {"nn_model." + k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        x32 = O.ddpm_sample(PD, sched, n_T, 4, xT, zc, w, steps=steps).double().numpy()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            x16 = O.ddpm_sample(PD, sched, n_T, 4, xT, zc, w, steps=steps).float().double().numpy()
    x0 = xT.double().numpy()
    return {"n": n, "steps": steps, "guide_w": w,
            "x_mse_vs_oracle32": float(((xg - x32) ** 2).mean()), "oracle_autocast_bf16_mse": float(((x16 - x32) ** 2).mean()),
            "x_maxabs_vs_oracle32": float(np.abs(xg - x32).max()), "oracle_autocast_bf16_maxabs": float(np.abs(x16 - x32).max()),
            "moved_power": float(((x32 - x0) ** 2).mean()),
            "graph_vs_eager_injected_maxabs": float(np.abs(xg - xe).max())}


def train3_case(dtype):
    """Three optimiser steps (accumulation 2, clip 1.0, AdamW lr 1e-3 wd 1e-2) through the product path — DDPM.forward / ACCUM,
    backward, FusedAdamW.step — on the draws of tests/golden/train3.npz (the reference's loop on the CPU, fp32)."""
    import diffusionmodel_amd as D
    g = npz("train3")
    lr, wd, accum, n_opt, B = [float(v) for v in g["hyper"]]
    accum, n_opt, B = int(accum), int(n_opt), int(B)
    S, n_T = 64, 1000
    ddpm = D.DDPM(D.ContextUnet(3, 32, 4, bottleneck_k=4, dtype=dtype), (1e-4, 0.02), n_T, DEV, drop_prob=0.1)
    sd = {k: synth.synth_tensor(k, tuple(s)) for k, s in SCHEMA["ddpm_fwd64"]}
    for k in D.SCHEDULE_KEYS:
        sd[k] = D.ddpm_schedules(1e-4, 0.02, n_T)[k]
    ddpm.load_state_dict(sd)
    ddpm.train()
    opt = D.FusedAdamW(ddpm.parameters(), lr=lr, weight_decay=wd, max_grad_norm=1.0,
                       shadow_dtype=dtype if dtype != torch.float32 else torch.bfloat16)
    losses, norms = [], []
    opt.zero_grad()
    for m in range(accum * n_opt):
        tag = f"train3.m{m}"
        x = si(tag + ".x", (B, 3, S, S)).to(DEV)
        c = torch.tensor([(m + i) % 4 for i in range(B)]).to(DEV)
        am = synth.synth_attn_mask(B, S).to(DEV)
        noise = synth.synth_noise(tag + ".noise", (B, 3, S, S)).to(DEV)
        loss = ddpm(x, c, am, ts=torch.tensor(g[f"ts.{m}"]).to(DEV), noise=noise, ctx_mask=torch.tensor(g[f"keep.{m}"]).to(DEV)) / accum
        losses.append(float(loss.item()) * accum)
        ddpm.scaler.scale(loss).backward()                                               # new_scripy.py:792
        if (m + 1) % accum == 0:
            ddpm.scaler.unscale_(opt)
            ddpm.scaler.step(opt)                                                          # :797-801
            ddpm.scaler.update()
            norms.append(float(opt.grad_norm().item()) / _loss_scale(dtype))
            opt.zero_grad()
    named = dict(ddpm.named_parameters())
    rec = {"losses": losses, "losses_ref": [float(v) for v in g["losses"]], "grad_norms": norms, "grad_norms_ref": [float(v) for v in g["grad_norms"]]}
    gb = npz("train3_bf16")                    # the reference's loop under autocast(bfloat16): how far bf16 moves the reference itself
    rec["losses_ref_autocast_bf16"], rec["grad_norms_ref_autocast_bf16"] = [float(v) for v in gb["losses"]], [float(v) for v in gb["grad_norms"]]
    rec["ref_autocast_loss_rel_dev"] = [abs(a - b) / abs(b) for a, b in zip(rec["losses_ref_autocast_bf16"], rec["losses_ref"])]
    rec["ref_autocast_grad_norm_rel_dev"] = [abs(a - b) / abs(b) for a, b in zip(rec["grad_norms_ref_autocast_bf16"], rec["grad_norms_ref"])]
    rec["loss_rel_err"] = [abs(a - b) / abs(b) for a, b in zip(losses, rec["losses_ref"])]
    rec["grad_norm_rel_err"] = [abs(a - b) / abs(b) for a, b in zip(norms, rec["grad_norms_ref"])]
    pn_err = {}
    for cn, ch in ddpm.nn_model.named_children():
        ps = [p.detach().double().reshape(-1) for p in ch.parameters()]
        if ps and f"pnorm.{cn}" in g.files:
            pn_err[cn] = float(torch.cat(ps).norm().item() / float(g[f"pnorm.{cn}"]) - 1.0)
    rec["param_norm_rel_err"] = pn_err
    osd = opt.state_dict()
    idx = {n: i for i, (n, _) in enumerate(ddpm.named_parameters())}
    tens = {}
    for key in g.files:
        if key.startswith("p."):
            pn = key[2:]
            ref_p, ref_m, ref_v = g[key], g["m." + pn], g["v." + pn]
            st = osd["state"][idx[pn]]
            tens[pn] = {"param_maxabs_err": float(np.abs(named[pn].detach().cpu().numpy() - ref_p).max()),
                        "param_update_scale": lr * n_opt, "param_moved_maxabs": float(np.abs(ref_p - synth.synth_tensor(pn, tuple(ref_p.shape)).numpy()).max()),
                        "exp_avg_rel_err": float(np.abs(st["exp_avg"].cpu().numpy() - ref_m).max() / (np.abs(ref_m).max() + 1e-30)),
                        "exp_avg_sq_rel_err": float(np.abs(st["exp_avg_sq"].cpu().numpy() - ref_v).max() / (np.abs(ref_v).max() + 1e-30)),
                        "step": float(st["step"]), "step_ref": float(g["step." + pn])}
    rec["tensors"] = tens
    sd2 = ddpm.state_dict()
    rec["bn_buffers_rel_err"] = {k[4:]: float(np.abs(sd2[k[4:]].cpu().numpy() - g[k]).max() / (np.abs(g[k]).max() + 1e-30)) for k in g.files if k.startswith("buf.")}
    rec["num_batches_tracked"] = [int(sd2["nn_model.init_conv.conv1.1.num_batches_tracked"]), int(g["nbt"])]
    return rec


def bn_bwd_kernel_case():
    """conv3x3 + train-mode BatchNorm + GELU (one ConvBnAct node, halo kernel + BN kernels) in bf16 against torch fp32 on the CPU,
    next to the same torch modules run under autocast(bfloat16): the kernel-level yardstick for test_gpu_kernels."""
    import torch.nn.functional as F
    from diffusionmodel_amd import ops as o
    torch.manual_seed(0)
    B, Ci, Co, H = 2, 64, 128, 32
    x = torch.randn(B, Ci, H, H).bfloat16().float()
    conv_r = torch.nn.Conv2d(Ci, Co, 3, 1, 1)
    with torch.no_grad():
        conv_r.weight.copy_(conv_r.weight.bfloat16().float())
    bn_r = torch.nn.BatchNorm2d(Co)
    with torch.no_grad():
        bn_r.weight.uniform_(0.5, 1.5)
        bn_r.bias.uniform_(-0.3, 0.3)
    st = {k: v.clone() for k, v in bn_r.state_dict().items()}
    probe = torch.randn(B, Co, H, H)

    def torch_run(autocast):
        bn_r.load_state_dict(st)
        xr = x.clone().requires_grad_(True)
        conv_r.zero_grad(); bn_r.zero_grad()
        if autocast:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                y = F.gelu(bn_r(conv_r(xr)))
        else:
            y = F.gelu(bn_r(conv_r(xr)))
        (y.float() * probe).sum().backward()
        return y.detach().float(), xr.grad.clone(), conv_r.weight.grad.clone(), bn_r.weight.grad.clone(), bn_r.bias.grad.clone()
    ref = torch_run(False)
    a16 = torch_run(True)

    class Holder:
        def __init__(s, w, b):
            s.weight = torch.nn.Parameter(w.to(DEV).contiguous(memory_format=torch.channels_last))
            s.bias = torch.nn.Parameter(b.to(DEV))
    bn_d = torch.nn.BatchNorm2d(Co).to(DEV)
    bn_d.load_state_dict(st)
    conv_d = Holder(conv_r.weight.detach().clone(), conv_r.bias.detach().clone())
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV).bfloat16().requires_grad_(True)
    y = o.conv_bn_act(xd, None, conv_d, bn_d, o.ConvSpec(3, 3, 1, 1, o.ACT_GELU, bn_d))
    (y.float() * probe.permute(0, 2, 3, 1).contiguous().to(DEV)).sum().backward()
    hip = (y.detach().float().permute(0, 3, 1, 2).cpu(), xd.grad.float().permute(0, 3, 1, 2).cpu(), conv_d.weight.grad.cpu(),
           bn_d.weight.grad.cpu(), bn_d.bias.grad.cpu())

    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / b.double().abs().max())

    def rms(a, b):
        return float(((a.double() - b.double()) ** 2).mean().sqrt() / (b.double() ** 2).mean().sqrt())
    names = ("y", "dx", "dw", "dgamma", "dbeta")
    return {n: {"hip_maxrel": rel(h, r), "torch_autocast_maxrel": rel(a, r), "hip_rms_rel": rms(h, r), "torch_autocast_rms_rel": rms(a, r)}
            for n, h, a, r in zip(names, hip, a16, ref)}


def measure_all():
    out = {"unet32_64": {"bf16": unet_case(torch.bfloat16), "fp16": unet_case(torch.float16), "fp32": unet_case(torch.float32)},
           "ddpm_fwd64": {"bf16": ddpm_case(torch.bfloat16), "fp16": ddpm_case(torch.float16), "fp32": ddpm_case(torch.float32)},
           "f128_b8_bf16_vs_oracle_fp32": f128_case(),
           "f128_b2_vs_reference": {"fp32": f128_b2_case(torch.float32), "bf16": f128_b2_case(torch.bfloat16)},
           "f128_b2_band_bf16_vs_reference": f128_b2_band_case(),
           "cfg2_full_size_b64_bf16_vs_oracle_fp32": cfg2_full_size_case(),
           "cfg4_sample_f128_n16_bf16_graph_vs_oracle_fp32": sample_f128_case(),
           "train3": {"fp32": train3_case(torch.float32), "bf16": train3_case(torch.bfloat16), "fp16": train3_case(torch.float16)},
           "conv_bn_gelu_kernel_bf16": bn_bwd_kernel_case()}
    return out
