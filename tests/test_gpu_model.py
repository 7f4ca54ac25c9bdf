"""Model-level parity on a real MI355X: the HIP class surface against (a) the golden fixtures generated
from the imported reference (tests/golden/, committed) and (b) the CPU oracle on the same seeded inputs.

Stated tolerances (fp32 mode): eps-prediction max-abs <= 1e-4 in eval mode; train mode is limited by the
reference's own fp32 rounding through BatchNorm over tiny batches (its fp32-vs-fp64 noise on these
fixtures is 4e-5..9e-5, see make_golden.py) and is held to the same 1e-4 against the fp64 reference
output since r03.  bf16 mode: MSE-based (bf16 autocast vs fp32 of the reference is 5e-6 MSE in eval)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import synth, unet_ref as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = os.path.join(os.path.dirname(__file__), "golden")
SCHEMA = json.load(open(os.path.join(G, "schema.json")))
si = synth.synth_input


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def npz(name):
    return np.load(os.path.join(G, name + ".npz"))


def load_synth(mod, tag, prefix=""):
    sd = {k: synth.synth_tensor(prefix + k, tuple(s)) for k, s in SCHEMA[tag]}
    mod.load_state_dict(sd, strict=True)
    return mod.to(DEV)


def maxerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def relerr(a, b):
    b = np.asarray(b, np.float64)
    return maxerr(a, b) / max(float(np.abs(b).max()), 1e-12)


def run_block(tag, mod, inputs, tol=1e-4, gtol=2e-3):
    g = npz(tag)
    for train in (False, True):
        load_synth(mod, tag)
        mod.train(train)
        mode = "train" if train else "eval"
        ins = [(v.clone().to(DEV).requires_grad_(True) if (v.is_floating_point() and k != "mask") else v.to(DEV)) for k, v in inputs.items()]
        y = mod(*ins)
        assert relerr(y.detach().cpu().numpy(), g[f"{mode}.y"]) < tol, (tag, mode, "y")
        probe = si(tag + ".probe", tuple(y.shape)).to(DEV)
        mod.zero_grad()
        (y * probe).sum().backward()
        for (k, _), v in zip(inputs.items(), ins):
            if f"{mode}.d_{k}" in g.files:
                assert relerr(v.grad.cpu().numpy(), g[f"{mode}.d_{k}"]) < gtol, (tag, mode, "d_" + k)
        for k, p in mod.named_parameters():
            ref = g[f"{mode}.g.{k}"]
            got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
            if np.abs(ref).max() < 5e-5 and k.endswith(".bias"):   # conv bias in front of train-mode BatchNorm: mathematically zero, fp32 noise on both sides
                assert np.abs(got).max() < 1e-4, (tag, mode, k)
            else:
                assert relerr(got, ref) < gtol, (tag, mode, k, relerr(got, ref))
        if train:
            sd = mod.state_dict()
            for k in g.files:
                if k.startswith("train.buf."):
                    assert relerr(sd[k[len("train.buf."):]].cpu().numpy(), g[k]) < 1e-5, (tag, k)


def test_se_block():
    import diffusionmodel_amd as D
    run_block("se32", D.SEBlock(32), {"x": si("se32.x", (2, 32, 8, 8))})


@pytest.mark.parametrize("tag,shape", [("ca32_8", (3, 32, 8, 8)), ("ca32_16", (2, 32, 16, 16))])
def test_coord_attn(tag, shape):
    import diffusionmodel_amd as D
    run_block(tag, D.CoordAttn(32), {"x": si(tag + ".x", shape)})


@pytest.mark.parametrize("tag,cin,res", [("rcb_3_16_res", 3, True), ("rcb_16_16_res", 16, True), ("rcb_16_16_plain", 16, False)])
def test_res_conv_block(tag, cin, res):
    import diffusionmodel_amd as D
    g = npz(tag)
    mod = D.ResConvBlock(cin, 16, res)
    if cin == 3:   # gradient w.r.t. the channel-padded stem input is not provided (never needed by the net)
        for train in (False, True):
            load_synth(mod, tag)
            mod.train(train)
            mode = "train" if train else "eval"
            y = mod(si(tag + ".x", (2, 3, 16, 16)).to(DEV))
            assert relerr(y.detach().cpu().numpy(), g[f"{mode}.y"]) < 1e-4
            (y * si(tag + ".probe", tuple(y.shape)).to(DEV)).sum().backward()
            for k, p in mod.named_parameters():
                ref = g[f"{mode}.g.{k}"]
                if not (np.abs(ref).max() < 5e-5 and k.endswith(".bias")):
                    assert relerr(p.grad.cpu().numpy(), ref) < 2e-3, (k, mode)
            mod.zero_grad()
        return
    run_block(tag, mod, {"x": si(tag + ".x", (2, cin, 16, 16))})


def test_unet_down():
    import diffusionmodel_amd as D
    run_block("down_32_64", D.UnetDown(32, 64), {"x": si("down_32_64.x", (2, 32, 16, 16))})


def test_unet_up():
    import diffusionmodel_amd as D
    run_block("up_64_16", D.UnetUp(64, 16), {"x": si("up_64_16.x", (2, 32, 8, 8)), "skip": si("up_64_16.skip", (2, 32, 8, 8))})


@pytest.mark.parametrize("tag,d", [("fc_1_32", 1), ("fc_4_32", 4)])
def test_embed_fc(tag, d):
    import diffusionmodel_amd as D
    run_block(tag, D.EmbedFC(d, 32), {"x": si(tag + ".x", (5, d))})


def test_local_enhancer_standalone():
    import diffusionmodel_amd as D
    run_block("le16", D.LocalEnhancer(16), {"x": si("le16.x", (2, 16, 16, 16)), "mask": synth.synth_attn_mask(2, 16)})


def _child_norms(net):
    d = {}
    for cn, ch in net.named_children():
        d[cn] = sum(float((p.grad.double() ** 2).sum()) for p in ch.parameters() if p.grad is not None) ** 0.5
    return d


_ORACLE_GRADS = {}


def _oracle_train_grads(tag, S, x, c, t, mk):
    """Train-mode parameter gradients of the CPU oracle in fp64 and fp32 (cached per fixture)."""
    if tag not in _ORACLE_GRADS:
        out = []
        for dt_ in (torch.float64, torch.float32):
            P = {}
            for k_, s_ in SCHEMA[tag]:
                v = synth.synth_tensor(k_, tuple(s_))
                if v.is_floating_point():
                    v = v.to(dt_)
                    if "running" not in k_:
                        v.requires_grad_(True)
                P[k_] = v
            eps = O.context_unet(P, x.to(dt_), c, t.to(dt_), mk.to(dt_), True)
            (eps * si(tag + ".probe", tuple(eps.shape)).to(dt_)).mean().backward()
            out.append({k_: v.grad.numpy() for k_, v in P.items() if v.is_floating_point() and v.grad is not None})
        _ORACLE_GRADS[tag] = out
    return _ORACLE_GRADS[tag]


@pytest.mark.parametrize("tag,S,k", [("unet32_64", 64, 4), ("unet32_128", 128, 8)])
def test_context_unet_vs_reference_fixture(tag, S, k):
    import diffusionmodel_amd as D
    g = npz(tag)
    x = si(tag + ".x", (2, 3, S, S))
    c, t, mk = torch.tensor(g["c"]), torch.tensor(g["t"]), torch.tensor(g["ctx_mask"])
    for train in (False, True):
        mode = "train" if train else "eval"
        net = load_synth(D.ContextUnet(3, 32, 4, bottleneck_k=k, dtype=torch.float32), tag)
        net.train(train)
        eps = net(x.to(DEV), c.to(DEV), t.to(DEV), mk.to(DEV))
        e = eps.detach().cpu().numpy()
        err32, err64 = maxerr(e, g[f"{mode}.eps"]), maxerr(e, g[f"{mode}.eps64"])
        print(f"{tag} {mode}: max|eps - ref32| = {err32:.2e}, max|eps - ref64| = {err64:.2e}")
        # north star: 1e-4 in fp32 mode, eval and train.  Measured (r03): 64x64 1.6e-6 / 4.6e-5, 128x128 2.6e-6 / 7.0e-5 (the reference's
        # own fp32-vs-fp64 noise in train mode at B = 2 is 2.3e-5)
        assert err64 < 1e-4
        probe = si(tag + ".probe", tuple(eps.shape)).to(DEV)
        loss = (eps * probe).mean()
        loss.backward()
        assert abs(loss.item() - float(g[f"{mode}.loss"])) < 1e-5
        norms = _child_norms(net)
        for key in g.files:
            if key.startswith(f"{mode}.gn."):
                cn = key.split(".", 2)[2]
                if cn == "local_enhance":
                    continue
                ref = float(g[key])
                assert abs(norms[cn] - ref) <= (3e-3 if train else 5e-4) * max(ref, 1e-3), (cn, norms[cn], ref)
            if key.startswith(f"{mode}.g."):
                pn = key.split(".", 2)[2]
                got = dict(net.named_parameters())[pn].grad.cpu().numpy()
                if not train:
                    # scalar gates (CoordAttn gamma / alpha) are cancellation-heavy sums of ~1e-8: their last digits follow
                    # the summation order of the dense layers (split-K), so they get a looser relative bar
                    assert relerr(got, g[key]) < (5e-3 if got.size == 1 else 5e-4), (pn, relerr(got, g[key]))
                else:
                    # train-mode BatchNorm over B=2 amplifies fp32 rounding: measure the CPU oracle's own
                    # fp32-vs-fp64 deviation for this parameter and allow a few times that
                    g64, g32 = _oracle_train_grads(tag, S, x, c, t, mk)
                    noise = relerr(g32[pn], g64[pn])
                    # scalar parameters (CoordAttn gamma/alpha) are sums with heavy cancellation: under 1e-7
                    # relative input noise the HIP value of ca2.gamma_h itself moves by 1.2e-2 and ca1.alpha by
                    # 3e-3 (scripts/probe_conditioning.py), so they get a conditioning-sized bar
                    if got.size == 1:   # |value| ~ 2e-5 here: relative 5e-2 plus an absolute floor of 5e-6
                        assert maxerr(got, g64[pn]) < 5e-2 * abs(float(g64[pn].ravel()[0])) + 5e-6, (pn, got, g64[pn])
                    else:
                        bar = max(5e-3, 4 * noise)
                        assert relerr(got, g64[pn]) < bar, (pn, relerr(got, g64[pn]), noise)
        if train:
            sd = net.state_dict()
            for key in g.files:
                if key.startswith("train.buf."):
                    assert relerr(sd[key[len("train.buf."):]].cpu().numpy(), g[key]) < 1e-4, key
            assert int(sd["init_conv.conv1.1.num_batches_tracked"]) == 1


def test_ddpm_forward_loss_and_sampling_vs_reference_fixture():
    import diffusionmodel_amd as D
    tag = "ddpm_fwd64"
    g = npz(tag)
    B, S, n_T = 4, 64, 1000
    x = si(tag + ".x", (B, 3, S, S)).to(DEV)
    c = torch.tensor([(i + 2) % 4 for i in range(B)]).to(DEV)
    am = synth.synth_attn_mask(B, S).to(DEV)
    ts, keep = torch.tensor(g["ts"]).to(DEV), torch.tensor(g["keep"]).to(DEV)
    noise = synth.synth_noise(tag + ".noise", (B, 3, S, S)).to(DEV)
    for train in (True, False):
        ddpm = D.DDPM(D.ContextUnet(3, 32, 4, bottleneck_k=4), (1e-4, 0.02), n_T, DEV, drop_prob=0.1)
        sd = {k: synth.synth_tensor(k, tuple(s)) for k, s in SCHEMA[tag]}
        for k in D.SCHEDULE_KEYS:
            sd[k] = D.ddpm_schedules(1e-4, 0.02, n_T)[k]
        ddpm.load_state_dict(sd)
        ddpm.train(train)
        loss = ddpm(x, c, am, ts=ts, noise=noise, ctx_mask=keep)
        mode = "train" if train else "eval"
        assert abs(loss.item() - float(g[f"{mode}.loss"])) < 2e-5, (loss.item(), float(g[f"{mode}.loss"]))
        if train:
            loss.backward()
            norms = _child_norms(ddpm.nn_model)
            for key in g.files:
                if key.startswith("train.gn.") and not key.endswith("local_enhance") and not key.endswith("to_vec"):
                    ref = float(g[key])
                    assert abs(norms[key.split(".", 2)[2]] - ref) <= 3e-3 * max(ref, 1e-3), key
    # sampling trajectories with injected noise, reference convention, with and without encoder de-dup
    for stag, T, n, w in (("sample64_T5", 5, 4, 2.0), ("sample64_T3_w0", 3, 8, 0.0)):
        gs = npz(stag)
        d2 = D.DDPM(D.ContextUnet(3, 32, 4, bottleneck_k=4), (1e-4, 0.02), T, DEV, drop_prob=0.0)
        sd = {k: synth.synth_tensor(k, tuple(s)) for k, s in SCHEMA[tag]}
        for k in D.SCHEDULE_KEYS:
            sd[k] = D.ddpm_schedules(1e-4, 0.02, T)[k]
        d2.load_state_dict(sd)
        d2.eval()
        x_T = synth.synth_noise(f"{stag}.z0", (n, 3, 64, 64))
        zs = [synth.synth_noise(f"{stag}.z{j + 1}", (n, 3, 64, 64)) for j in range(T)]
        for dedup in (True, False):
            xs = d2.sample(n, (3, 64, 64), DEV, guide_w=w, x_T=x_T, zs=zs, dedup=dedup)
            err = maxerr(xs.cpu().numpy(), gs["x"])
            print(f"{stag} dedup={dedup}: max|x - ref| = {err:.2e}")
            assert err < 1e-4 * T + 1e-5, (stag, dedup, err)


def test_sampling_graph_replay_equals_eager():
    import diffusionmodel_amd as D
    T, n = 6, 4
    d = D.DDPM(D.ContextUnet(3, 32, 4, bottleneck_k=4), (1e-4, 0.02), T, DEV, drop_prob=0.0)
    sd = {k: synth.synth_tensor(k, tuple(s)) for k, s in SCHEMA["ddpm_fwd64"]}
    for k in D.SCHEDULE_KEYS:
        sd[k] = D.ddpm_schedules(1e-4, 0.02, T)[k]
    d.load_state_dict(sd)
    d.eval()
    x_T = synth.synth_noise("g.z0", (n, 3, 64, 64))
    a = d.sample(n, (3, 64, 64), DEV, guide_w=4.0, x_T=x_T, seed=77, use_graph=False)
    b = d.sample(n, (3, 64, 64), DEV, guide_w=4.0, x_T=x_T, seed=77, use_graph=True)
    assert torch.equal(a, b)
    assert torch.isfinite(a).all()


def test_mnist_net_vs_reference_fixture():
    from diffusionmodel_amd import mnist as DM
    g = npz("mnist16")
    x = si("mnist16.x", (3, 1, 28, 28)).to(DEV)
    c, t, mk = torch.tensor([1, 7, 4]).to(DEV), torch.tensor([0.2, 0.55, 0.9]).to(DEV), torch.tensor([0.0, 1.0, 0.0]).to(DEV)
    for train in (False, True):
        mode = "train" if train else "eval"
        net = load_synth(DM.ContextUnet(1, 16, 10), "mnist16")
        net.train(train)
        eps = net(x, c, t, mk)
        assert maxerr(eps.detach().cpu().numpy(), g[f"{mode}.eps"]) < (3e-4 if train else 1e-4)
        (eps * si("mnist16.probe", tuple(eps.shape)).to(DEV)).mean().backward()
        norms = _child_norms(net)
        for key in g.files:
            if key.startswith(f"{mode}.gn."):
                ref = float(g[key])
                assert abs(norms.get(key.split(".", 2)[2], 0.0) - ref) <= (3e-3 if train else 5e-4) * max(ref, 1e-3), key
    net = load_synth(DM.ContextUnet(1, 16, 10), "mnist16")
    ddpm = DM.DDPM(net, (1e-4, 0.02), 400, DEV, drop_prob=0.1)
    ddpm.train()
    ts = torch.tensor([1 + (313 * i + 96) % 400 for i in range(3)]).to(DEV)
    noise = synth.synth_noise("mnist16.noise", (3, 1, 28, 28)).to(DEV)
    drop = torch.tensor([float(i % 3 == 1) for i in range(3)]).to(DEV)
    loss = ddpm(x, c, ts=ts, noise=noise, context_mask=drop)
    assert abs(loss.item() - float(g["ddpm.loss"])) < 2e-5
    net = load_synth(DM.ContextUnet(1, 16, 10), "mnist16")
    d2 = DM.DDPM(net, (1e-4, 0.02), 4, DEV, drop_prob=0.1)
    d2.eval()
    x_T = synth.synth_noise("mnist16.s.z0", (10, 1, 28, 28))
    zs = [synth.synth_noise(f"mnist16.s.z{j + 1}", (10, 1, 28, 28)) for j in range(4)]
    xs, store = d2.sample(10, (1, 28, 28), DEV, guide_w=0.5, x_T=x_T, zs=zs)
    assert maxerr(xs.cpu().numpy(), g["sample.x"]) < 5e-4
    assert tuple(store.shape) == tuple(g["sample.store_shape"])


def test_bf16_mode_against_fp32_oracle():
    """bf16 throughput mode: eps MSE vs the fp32 oracle on the same weights (eval), and a train step that
    produces finite gradients whose direction agrees with the fp32 ones."""
    import diffusionmodel_amd as D
    tag = "unet32_64"
    g = npz(tag)
    x = si(tag + ".x", (2, 3, 64, 64))
    c, t, mk = torch.tensor(g["c"]), torch.tensor(g["t"]), torch.tensor(g["ctx_mask"])
    net = load_synth(D.ContextUnet(3, 32, 4, bottleneck_k=4, dtype=torch.bfloat16), tag)
    net.eval()
    with torch.no_grad():
        eps = net(x.to(DEV), c.to(DEV), t.to(DEV), mk.to(DEV)).cpu().numpy()
    mse = float(((eps - g["eval.eps64"]) ** 2).mean())
    ref_pow = float((g["eval.eps64"] ** 2).mean())
    print(f"bf16 eval eps MSE {mse:.3e} (signal power {ref_pow:.3e}), max abs {maxerr(eps, g['eval.eps64']):.3e}")
    assert mse < 2e-4 * max(ref_pow, 1.0)
    net.train()
    e2 = net(x.to(DEV), c.to(DEV), t.to(DEV), mk.to(DEV))
    probe = si(tag + ".probe", tuple(e2.shape)).to(DEV)
    (e2 * probe).mean().backward()
    ref = g["train.g.out.3.weight"]
    got = net.out[3].weight.grad.cpu().numpy()
    assert np.isfinite(got).all()
    cos = float((got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30))
    print(f"bf16 train: cosine(out.3 wgrad, fp32 reference) = {cos:.5f}")
    assert cos > 0.99


def test_full_size_properties_cfg2():
    """BASELINE cfg-2 size (64x64, F=128, B=64, bf16): size-independent properties — eval forward is
    batch-separable (sample i of a batch == the same sample alone), CFG de-dup == naive doubled batch,
    train step yields finite loss/grads and the fused optimiser moves every parameter."""
    import diffusionmodel_amd as D
    torch.manual_seed(0)
    net = D.ContextUnet(3, 128, 4, bottleneck_k=4, dtype=torch.bfloat16).to(DEV)
    ddpm = D.DDPM(net, (1e-4, 0.02), 1000, DEV, drop_prob=0.1)
    B = 64
    x = torch.randn(B, 3, 64, 64, device=DEV).clamp(-1, 1)
    c = torch.randint(0, 4, (B,), device=DEV)
    t = torch.rand(B, device=DEV)
    mk = torch.ones(B, device=DEV)
    net.eval()
    with torch.no_grad():
        full = net(x, c, t, mk)
        one = net(x[5:6], c[5:6], t[5:6], mk[5:6])
    assert torch.isfinite(full).all()
    assert (full[5:6] - one).abs().max().item() < 0.05 * full.abs().max().item() + 1e-3
    xs1 = ddpm.sample(8, (3, 64, 64), DEV, guide_w=2.0, seed=5, steps=2, dedup=True)
    xs2 = ddpm.sample(8, (3, 64, 64), DEV, guide_w=2.0, seed=5, steps=2, dedup=False)
    assert (xs1 - xs2).abs().max().item() < 0.05 * xs2.abs().max().item()
    net.train()
    opt = D.FusedAdamW(ddpm.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0)
    am = torch.full((B, 64, 64), 0.5, device=DEV)
    am[:, 32:] = 1.0
    am[:, 10:20, 10:30] = 3.0
    before = opt.flat_p.clone()
    opt.zero_grad()
    loss = ddpm(x, c, am)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    assert math_isfinite(loss.item())
    assert torch.isfinite(opt.flat_g).all() and torch.isfinite(opt.flat_p).all()
    moved = (opt.flat_p != before).float().mean().item()
    assert moved > 0.9, moved


def math_isfinite(v):
    import math
    return math.isfinite(v)


@pytest.mark.parametrize("mode", ["graph", "plan"])
def test_graphed_train_step_matches_eager_and_leaves_state_alone(mode):
    """GraphedTrainStep (mode="graph": hipGraph replay; mode="plan": the captured nodes re-issued from C as plain launches,
    include/dm_amd.h dm_plan_*): (1) warm-up + capture must not change weights, optimiser moments, BatchNorm buffers or RNG
    counters; (2) replays must train exactly like eager steps.  n_T = 1 and drop_prob = 0 make the torch-RNG draws
    (timesteps, context mask) deterministic, the DDPM noise comes from the Philox kernel's device-side offset in
    both modes, so the two trajectories differ only by fp32 atomic ordering in the small weight-gradient launches."""
    import diffusionmodel_amd as D

    def make():
        torch.manual_seed(7)
        net = D.ContextUnet(3, 32, 4, bottleneck_k=4, dtype=torch.float32)
        ddpm = D.DDPM(net, (1e-4, 0.02), 1, DEV, drop_prob=0.0)
        ddpm.train()
        ddpm.rng_seed = 99
        opt = D.FusedAdamW(ddpm.parameters(), lr=1e-3, weight_decay=1e-5, max_grad_norm=1.0)
        return ddpm, opt

    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 64, 64, generator=g).to(DEV)
    c = torch.randint(0, 4, (2,), generator=g).to(DEV)
    am = torch.ones(2, 64, 64).to(DEV)

    da, oa = make()
    losses_a = []
    for _ in range(3):
        oa.zero_grad()
        loss = da(x, c, am)
        loss.backward()
        oa.step()
        losses_a.append(loss.item())

    db, ob = make()
    p0, m0 = ob.flat_p.clone(), ob.exp_avg.clone()
    bufs0 = [b.clone() for b in db.buffers()]
    step = D.GraphedTrainStep(db, ob, x, c, am, mode=mode)
    if mode == "plan":
        pl = step.plan
        assert pl.n_kernels > 300 and pl.n_segments == 1 and pl.n_markers == 0 and pl.n_ops == pl.n_kernels + pl.n_memsets
        names = [nm for kind, nm in pl.op_names() if kind == 0]
        assert any("adamw_kernel" in nm for nm in names) and any("conv" in nm for nm in names)
        foreign = pl.foreign_kernels()                      # torch-issued kernels left in the step: fills / the step count / index plumbing
        # (two optimisers are alive in this test, so the gradient scratch arena is off and every small accumulator is a torch.zeros
        #  fill; with one optimiser — bench.py — the step holds two fills)
        assert sum(n for nm, n in foreign.items() if "FillFunctor" not in nm) <= 10, foreign
        assert not any("random" in nm or "bernoulli" in nm or "CUDAFunctor_add" in nm and "BFloat16" in nm for nm in foreign), foreign
    assert torch.equal(ob.flat_p, p0) and torch.equal(ob.exp_avg, m0) and ob._step == 0 and int(ob._step_dev.item()) == 0
    assert all(torch.equal(b, b0) for b, b0 in zip(db.buffers(), bufs0))
    assert db._rng_calls == 0 and int(db._rng_dev.item()) == 0
    losses_b = [step(x, c, am).item() for _ in range(3)]
    assert ob._step == 3 and int(ob._step_dev.item()) == 3 and int(db._rng_dev.item()) == 3
    for la, lb in zip(losses_a, losses_b):
        assert abs(la - lb) <= 2e-4 * max(abs(la), 1e-3), (losses_a, losses_b)
    assert losses_b[-1] < losses_b[0]
    # Adam turns noise-level gradients (conv biases in front of train-mode BatchNorm are mathematically zero) into +-lr moves
    # whose sign follows the fp32 atomic order, so the weights agree to a few lr, not to rounding
    rel = ((oa.flat_p - ob.flat_p).norm() / oa.flat_p.norm()).item()
    assert rel < 5e-3, rel
    # BatchNorm bookkeeping of the replays reaches the state dict
    nbt_a = [v for k, v in da.state_dict().items() if k.endswith("num_batches_tracked")]
    nbt_b = [v for k, v in db.state_dict().items() if k.endswith("num_batches_tracked")]
    assert nbt_a and all(int(a) == int(b) for a, b in zip(nbt_a, nbt_b))
    # an eager step after the replays still works (packs are rebuilt lazily)
    ob.zero_grad()
    loss = db(x, c, am)
    loss.backward()
    ob.step()
    assert torch.isfinite(loss).item()


def test_train_engine_plan_then_eager_tail_batch_starts_from_a_zero_gradient():
    """a replayed plan zeroes the flat gradient buffer only at its start, so the buffer still holds the step's
    gradient afterwards; the eager step that follows (the short last batch of an epoch — DataLoader(drop_last=False),
    new_scripy.py:700-707 — or injected draws) must not accumulate on top of it.  TrainEngine(use_plan=True) over
    full, full, SHORT, full, full micro-batches == the all-eager engine: per-step loss and pre-clip gradient norm."""
    import diffusionmodel_amd as D
    from diffusionmodel_amd.train import TrainEngine

    def make(use_plan):
        torch.manual_seed(7)
        net = D.ContextUnet(3, 32, 4, bottleneck_k=4, dtype=torch.float32)
        ddpm = D.DDPM(net, (1e-4, 0.02), 1, DEV, drop_prob=0.0)
        ddpm.train()
        ddpm.rng_seed = 99
        opt = D.FusedAdamW(ddpm.parameters(), lr=1e-3, weight_decay=1e-5, max_grad_norm=1.0)
        return ddpm, opt, TrainEngine(ddpm, opt, accum_steps=1, use_plan=use_plan)

    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 64, 64, generator=g).to(DEV)
    c = torch.randint(0, 4, (2,), generator=g).to(DEV)
    am = torch.ones(2, 64, 64).to(DEV)
    sizes = [2, 2, 1, 2, 2]

    def run(use_plan):
        ddpm, opt, eng = make(use_plan)
        rec = []
        for i, n in enumerate(sizes):
            loss = eng.micro_batch(x[:n], c[:n], am[:n], last_in_epoch=(i == 2))
            rec.append((loss.item(), opt.grad_norm().item()))
        return rec, eng, opt

    ra, ea, oa = run(False)
    rb, eb, ob = run(True)
    assert ea.plan is None and eb.plan is not None and eb._planned.replays == 4 and eb.opt_steps == ea.opt_steps == 5
    assert ob._step == oa._step == 5 and int(ob._step_dev.item()) == 5
    for i, ((la, na), (lb, nb)) in enumerate(zip(ra, rb)):
        # through the step after the tail the two runs agree to the atomics band; after five lr = 1e-3 Adam steps on B = 2 they drift (1e-3)
        # (run-to-run band of this B = 2 / B = 1 trajectory, measured over the round's runs: the tail step's norm 1.7185 .. 1.7496 for the SAME
        #  engine — fp32 atomics through lr = 1e-3 Adam steps; the stale gradient of the bug doubles it)
        assert abs(la - lb) <= (2e-4 if i <= 2 else 5e-3) * max(abs(la), 1e-3), (ra, rb)
        assert abs(na - nb) <= (5e-2 if i <= 3 else 0.15) * na, (ra, rb)              # (the stale gradient makes the tail step's norm ~2x; Adam steps on atomics-ordered sums drift by ~5e-3)
    rel = ((oa.flat_p - ob.flat_p).norm() / oa.flat_p.norm()).item()
    assert rel < 5e-3, rel


def test_overlapped_gradient_reducer_two_ranks_on_one_gpu():
    """scripts/dp_rehearsal.py: two gloo ranks sharing cuda:0 run the real backward pass through OverlappedGradReducer
    (observation step + two overlapped steps) and compare the reduced flat gradient with a blocking host-side sum.
    The script pins the <= 64-KiB-LDS kernel variants: workgroups with more LDS do not survive preemption between
    processes that time-share one GPU on this stack (multi-GPU section)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29700 + os.getpid() % 200
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "scripts", "dp_rehearsal.py")],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "step 2" in r.stdout and "4/4" in r.stdout, r.stdout[-1500:]
    assert "planned DP:" in r.stdout, r.stdout[-1500:]               # the plan-replayed data-parallel step ran and agreed with the eager one


def test_dp_product_configuration_one_process_nccl_side_stream_load():
    """ONE process, nccl (RCCL) world size 1, the default 156-KiB-LDS kernels, OverlappedGradReducer in eager
    and launch-plan mode, with a streaming load on the reducer's side stream while the backward pass runs (tests/dp_nccl1_sidestream.py):
    the reduced flat gradient equals the no-reducer run of the same seeded pass to the fp32-atomics band.
    (The child runs with DM_DEVICE_GUARD=0: this pytest process holds the device claim but is idle while it waits.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DM_DEVICE_GUARD="0", MASTER_PORT=str(29500 + os.getpid() % 200))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "dp_nccl1_sidestream.py")], capture_output=True, text=True, timeout=900,
                       cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "dp_nccl1_sidestream OK" in r.stdout and "eager rep 5" in r.stdout, r.stdout[-2000:]


def test_shared_device_guard_selects_the_small_kernels_on_the_device():
    """after guard_shared_device / the per-process guard engage, the library really is on conv / wgrad variant 2 and a
    3x3 convolution then runs on a <= 64-KiB kernel (dm_last_conv_path != halo); restored afterwards."""
    from diffusionmodel_amd import _lib as L, ops as o
    lib = L.load()
    saved = dict(L._guard), L.LOCK_DIR
    import tempfile
    try:
        x = torch.randn(4, 32, 32, 64, device=DEV).bfloat16()
        w = torch.nn.Parameter(torch.randn(64, 64, 3, 3, device=DEV).contiguous(memory_format=torch.channels_last))

        class H:
            weight, bias = w, None
        sp = o.ConvSpec(3, 3, 1, 1)
        with torch.no_grad():
            y_big = o.conv_bn_act(x, None, H, None, sp)
        assert lib.dm_last_conv_path() == 1                                       # the halo-resident kernel (156 KiB of LDS)
        with tempfile.TemporaryDirectory() as td:
            L._guard.update(fd=None, path=None, shared=False, checked=0)
            L.LOCK_DIR = td
            assert L.device_guard(identity="dup") is False
            import fcntl
            fd2 = os.open(L._guard["path"], os.O_RDWR)                            # "another process" on the same device
            fcntl.flock(fd2, fcntl.LOCK_SH)
            assert L.device_guard(recheck=True) is True
            os.close(fd2)
            assert lib.dm_get_conv_variant() == 2
            with torch.no_grad():
                y_small = o.conv_bn_act(x, None, H, None, sp)
            assert lib.dm_last_conv_path() != 1
            assert torch.allclose(y_small.float(), y_big.float(), rtol=2e-2, atol=2e-2)
    finally:
        if L._guard["fd"] is not None:
            os.close(L._guard["fd"])
        L._guard.clear()
        L._guard.update(saved[0])
        L.LOCK_DIR = saved[1]
        lib.dm_set_conv_variant(L.DEFAULT_CONV_VARIANT)
        lib.dm_set_wgrad_variant(3)


def test_driver_engine_two_ranks_reproduce_the_reference_accumulation():
    """Row X2: the engine new_scripy.train_model drives (diffusionmodel_amd/train.py), as two data-parallel ranks (gloo, one GPU),
    reproduces the reference's single-process ACCUM_STEPS = 2 trajectory of tests/golden/train3.npz (new_scripy.py:777-803) within
    the fp32 bars of the single-process test: rank == micro-batch of the accumulation group."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29300 + os.getpid() % 200
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "tests", "dp_train3_ranks.py")],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "dp_train3 OK" in r.stdout, r.stdout[-1500:]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 0.25)])
def test_sample_shards_concatenate_to_single_call(dtype, tol):
    """samples [a, b) of the class-cycled batch, drawn with their slice of the Philox stream, equal
    rows [a, b) of the single call — what parallel.sample_sharded runs on each rank.  fp32: only the summation order of the
    batch-size-dependent split-K layers differs; bf16: a rounding flip in one layer propagates (loose bar, exact noise)."""
    import diffusionmodel_amd as D
    from diffusionmodel_amd import parallel
    torch.manual_seed(3)
    net = D.ContextUnet(3, 32, 4, bottleneck_k=4, dtype=dtype)
    ddpm = D.DDPM(net, (1e-4, 0.02), 50, DEV, drop_prob=0.1)
    ddpm.eval()
    kw = dict(guide_w=2.0, seed=77, steps=6)
    full = ddpm.sample(8, (3, 64, 64), DEV, **kw)
    lo = ddpm.sample(4, (3, 64, 64), DEV, first_sample=0, total_samples=8, **kw)
    hi = ddpm.sample(4, (3, 64, 64), DEV, first_sample=4, total_samples=8, use_graph=True, **kw)
    odd = ddpm.sample(2, (3, 64, 64), DEV, first_sample=5, total_samples=8, **kw)          # a shard need not start on a class boundary
    assert torch.isfinite(full).all() and full.std() > 0.1
    assert (torch.cat([lo, hi]) - full).abs().max().item() < tol
    assert (odd - full[5:7]).abs().max().item() < tol
    assert not torch.allclose(lo, hi)
    one = parallel.sample_sharded(ddpm, 8, (3, 64, 64), DEV, **kw)                          # no process group: one shard
    assert (one - full).abs().max().item() < tol
    with pytest.raises(D.DmError):
        ddpm.sample(4, (3, 64, 64), DEV, first_sample=6, total_samples=8, **kw)



def test_unseeded_sample_calls_draw_fresh_noise():
    """The reference draws fresh torch.randn noise on every sample() call (new_scripy.py:445,465): two unseeded calls differ, an
    explicit seed reproduces, and the sampler's stream is not the training-noise stream of the same instance."""
    import diffusionmodel_amd as D
    torch.manual_seed(3)
    ddpm = D.DDPM(D.ContextUnet(3, 32, 4, bottleneck_k=4, dtype=torch.float32), (1e-4, 0.02), 20, DEV, drop_prob=0.1)
    ddpm.eval()
    a = ddpm.sample(4, (3, 64, 64), DEV, guide_w=2.0, steps=2)
    b = ddpm.sample(4, (3, 64, 64), DEV, guide_w=2.0, steps=2)
    assert not torch.allclose(a, b) and (a - b).abs().mean().item() > 0.1
    c1 = ddpm.sample(4, (3, 64, 64), DEV, guide_w=2.0, steps=2, seed=11)
    c2 = ddpm.sample(4, (3, 64, 64), DEV, guide_w=2.0, steps=2, seed=11)
    assert torch.equal(c1, c2)
    # x_T of an unseeded call (0 steps) vs the first training-noise draw of the same instance (same shape): different numbers
    from diffusionmodel_amd import ops
    x_T = ddpm.sample(4, (3, 64, 64), DEV, guide_w=0.0, steps=0)
    train_noise = ops.randn((4, 3, 64, 64), DEV, ddpm._seed(), 1)
    assert not torch.allclose(x_T, train_noise)


def test_second_live_optimiser_switches_the_gradient_arena_off():
    """The pre-zeroed gradient scratch is recycled by FusedAdamW.step(); with two optimisers alive that would zero accumulators
    the other model has not consumed yet, so the arena serves exactly one optimiser."""
    import gc
    import diffusionmodel_amd as D
    from diffusionmodel_amd import ops
    gc.collect()
    n1 = D.ContextUnet(3, 32, 4, bottleneck_k=4, dtype=torch.float32).to(DEV)
    o1 = D.FusedAdamW(n1.parameters())
    first = ops.ARENA_ENABLED[0]
    n2 = D.ContextUnet(3, 32, 4, bottleneck_k=4, dtype=torch.float32).to(DEV)
    o2 = D.FusedAdamW(n2.parameters())
    assert ops.ARENA_ENABLED[0] is False
    # both models train side by side: backward of both, then both steps — gradients of the second must survive the first's step
    x = torch.randn(2, 3, 64, 64, device=DEV)
    c, t, mk = torch.tensor([0, 1], device=DEV), torch.tensor([0.3, 0.7], device=DEV), torch.ones(2, device=DEV)
    n1.train(); n2.train()
    n1(x, c, t, mk).mean().backward()
    n2(x, c, t, mk).square().mean().backward()
    g2 = n2.out[3].bias.grad.clone()
    o1.step()
    assert torch.equal(n2.out[3].bias.grad, g2) and g2.abs().sum().item() > 0
    o2.step()
    del o1, o2, n1, n2
    gc.collect()
    assert first in (True, False)


def test_replayed_forward_draws_fresh_timesteps_masks_and_noise():
    """A captured DDPM.forward replayed as a launch plan (no torch CUDAGraph.replay(), hence no help from torch's graph-safe RNG)
    must still draw new t / keep mask / noise on every replay: all three come from the library's Philox stream at a device-resident
    offset that the draw kernel itself advances (dm_draw_ts_keep, dm_randn_dev)."""
    import diffusionmodel_amd as D
    from diffusionmodel_amd.graph import LaunchPlan
    torch.manual_seed(0)
    ddpm = D.DDPM(D.ContextUnet(3, 32, 4, bottleneck_k=4, dtype=torch.float32), (1e-4, 0.02), 1000, DEV, drop_prob=0.5)
    ddpm.eval()
    ddpm.rng_seed = 123
    x = torch.randn(8, 3, 64, 64, device=DEV).clamp(-1, 1)
    c = torch.randint(0, 4, (8,), device=DEV)
    am = torch.ones(8, 64, 64, device=DEV)
    with torch.no_grad():
        eager = [ddpm(x, c, am).item() for _ in range(3)]            # offsets 1, 2, 3
        assert len({round(v, 6) for v in eager}) == 3
        ddpm._rng_dev.zero_()
        ddpm._rng_calls = 0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ddpm(x, c, am)                                           # warm-up (offset 1)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ddpm._rng_dev.zero_()
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(g):
            loss = ddpm(x, c, am)
        plan = LaunchPlan(g)
        assert not any("random" in nm or "bernoulli" in nm for _, nm in plan.op_names())       # no torch RNG kernels in the step
        replayed = []
        for _ in range(3):
            plan.run()
            replayed.append(loss.item())
    assert int(ddpm._rng_dev.item()) == 3
    assert replayed == pytest.approx(eager, rel=1e-6)                # the same three draws as three eager calls: t, mask and noise all moved


@pytest.mark.parametrize("w", [4.0, 6.0])
def test_full_length_graph_trajectory_cfg4(w):
    """BASELINE configs[3]: the 1000-step reverse loop under hipGraph capture at guidance scales 4 and 6 — all 1000 steps, the
    replayed graph against eager launches of the same seeded in-kernel noise (bit-equal), finite and non-degenerate."""
    import diffusionmodel_amd as D
    T, n = 1000, 4
    d = D.DDPM(D.ContextUnet(3, 32, 4, bottleneck_k=4), (1e-4, 0.02), T, DEV, drop_prob=0.0)
    sd = {k: synth.synth_tensor(k, tuple(s)) for k, s in SCHEMA["ddpm_fwd64"]}
    for k in D.SCHEDULE_KEYS:
        sd[k] = D.ddpm_schedules(1e-4, 0.02, T)[k]
    d.load_state_dict(sd)
    d.eval()
    a = d.sample(n, (3, 64, 64), DEV, guide_w=w, seed=11, use_graph=False)
    b = d.sample(n, (3, 64, 64), DEV, guide_w=w, seed=11, use_graph=True)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    assert a.std().item() > 1e-3


def test_full_length_trajectory_against_the_oracle():
    """All 1000 steps of DDPM.sample (T = 1000, w = 4, n = 4, fp32) with injected noise against the CPU oracle on the same noise:
    the stated bar for a full fp32 trajectory is 1e-3 max-abs at a pixel scale of a few units."""
    import diffusionmodel_amd as D
    T, n, w = 1000, 4, 4.0
    d = D.DDPM(D.ContextUnet(3, 32, 4, bottleneck_k=4), (1e-4, 0.02), T, DEV, drop_prob=0.0)
    P = {k: synth.synth_tensor(k, tuple(s)) for k, s in SCHEMA["ddpm_fwd64"]}
    sched = O.ddpm_schedules(1e-4, 0.02, T)
    sd = dict(P)
    for k in D.SCHEDULE_KEYS:
        sd[k] = D.ddpm_schedules(1e-4, 0.02, T)[k]
    d.load_state_dict(sd)
    d.eval()
    g = torch.Generator().manual_seed(5)
    x_T = torch.randn(n, 3, 64, 64, generator=g)
    zs = torch.randn(T, n, 3, 64, 64, generator=g)
    xs = d.sample(n, (3, 64, 64), DEV, guide_w=w, x_T=x_T, zs=list(zs)).cpu()
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = O.ddpm_sample(P, sched, T, 4, x_T, list(zs), w)
    err = (xs - ref).abs().max().item()
    print(f"1000-step trajectory w={w}: max|x - oracle| = {err:.2e}, pixel std {ref.std().item():.2f}, max |x| {ref.abs().max().item():.1f}")
    assert torch.isfinite(xs).all()
    assert err < 1e-3 * max(1.0, ref.abs().max().item())


def test_bench_two_rank_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2 --backend gloo` on a one-GPU box: the self-launcher starts two ranks that share cuda:0 (the library's
    shared-device guard selects the <= 64-KiB-LDS kernels), the train step runs as a launch plan with the bucketed all-reduce
    between its segments, sampling is sharded over the ranks, rank 0 prints ONE JSON line with ranks = 2.  The N-rank code path of
    the driver's scaling run, end to end (its numbers mean nothing here)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--sample-steps", "2", "--no-cpu-baseline", "--batch", "16"], capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and len(out["devices"]) == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 32 and out["config"]["exec"] == "plan" and out["config"]["backend"] == "gloo"
    assert out["value"] > 0 and out["loss"] == out["loss"] and out["sample"]["n"] == 32 and "error" not in out["sample"]
    assert "ranks share 1 device" in r.stderr
    # r04: what the N > 1 line is read against — the data-parallel facts, the calibration block and the repeated regions
    dp = out["dp"]
    assert dp["ranks"] == 2 and dp["buckets"] >= 1 and dp["allreduce_bytes_per_step_per_rank"] > 4 * 1e6 and dp["tail_wait_ms"] >= 0.0
    assert out["repeats"]["n"] == 5 and len(out["repeats"]["ms_per_step_all"]) == 5
    assert out["calibration"]["conv_random_us"] > 0 and out["calibration"]["copy_1GiB_TBps"] > 0


def test_replay_after_an_eager_backward_without_a_step_starts_from_clean_accumulators():
    """the captured step uses slices of the process-wide zero arena as zeroed accumulators (BatchNorm slot sums, small
    gradients) and re-zeroes the arena only at its END.  An eager train-mode forward + backward that is NOT followed by opt.step()
    (a logging pass, a partial accumulation group) leaves sums in those slices; GraphedTrainStep.__call__ puts the arena back first.
    The same replay — same weights, moments, BatchNorm buffers and noise counter — must give the same loss and gradient with and
    without such a pass in front of it."""
    import diffusionmodel_amd as D
    from diffusionmodel_amd import ops

    torch.manual_seed(11)
    net = D.ContextUnet(3, 32, 4, bottleneck_k=4, dtype=torch.float32)
    ddpm = D.DDPM(net, (1e-4, 0.02), 1, DEV, drop_prob=0.0).train()
    ddpm.rng_seed = 5
    opt = D.FusedAdamW(ddpm.parameters(), lr=1e-3, weight_decay=1e-5, max_grad_norm=1.0)
    if not ops.ARENA_ENABLED[0]:
        pytest.skip("another optimiser of this process is still alive: the gradient arena is off")
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 3, 64, 64, generator=g).to(DEV)
    c = torch.randint(0, 4, (2,), generator=g).to(DEV)
    am = torch.ones(2, 64, 64).to(DEV)
    step = D.GraphedTrainStep(ddpm, opt, x, c, am, mode="plan")

    def snapshot():
        return dict(p=opt.flat_p.clone(), m=opt.exp_avg.clone(), v=opt.exp_avg_sq.clone(), t=opt._step_dev.clone(), rng=ddpm._rng_dev.clone(),
                    bufs=[b.clone() for b in ddpm.buffers()], step=opt._step, calls=ddpm._rng_calls, nbt=[sp.nbt_pending for sp in step._specs])

    def restore(s):
        with torch.no_grad():
            opt.flat_p.copy_(s["p"]); opt.exp_avg.copy_(s["m"]); opt.exp_avg_sq.copy_(s["v"]); opt._step_dev.copy_(s["t"])
            ddpm._rng_dev.copy_(s["rng"])
            for b, b0 in zip(ddpm.buffers(), s["bufs"]):
                b.copy_(b0)
        opt._step, ddpm._rng_calls = s["step"], s["calls"]
        for sp, n in zip(step._specs, s["nbt"]):
            sp.nbt_pending = n
        opt.refresh_shadow(); ops.bump_weight_epoch(); ops.refresh_packs()

    s0 = snapshot()
    loss1 = step().clone()
    g1 = opt.flat_g.clone()
    restore(s0)
    ddpm(x, c, am).backward()                       # eager, train mode, no optimiser step
    assert ops.ZERO_ARENA.off != 0                  # (what the test is about: the arena is in use)
    restore(s0)                                     # (the eager pass advanced the noise counter and the BatchNorm buffers)
    loss2 = step().clone()
    g2 = opt.flat_g.clone()
    rel = ((g1 - g2).norm() / g1.norm()).item()
    print(f"replay after a dangling eager backward: loss {loss1.item():.6f} / {loss2.item():.6f}, gradient rel diff {rel:.2e}")
    assert abs(loss1.item() - loss2.item()) <= 1e-5 * abs(loss1.item())
    assert rel <= 1e-4, rel                         # fp32 atomics order only (dirty accumulators: O(1))


def test_two_backward_passes_over_one_graph_match_or_fail_loudly():
    """forked tensors (skip / residual / attention forks) hand the later consumer's gradient to the earlier consumer's
    input-gradient kernel through a stash (ops.GradFork).  A second backward over the same graph (retain_graph=True) must either
    produce the accumulated gradient (2 x) or raise — never drop the stashed half silently."""
    import diffusionmodel_amd as D
    from diffusionmodel_amd._lib import DmError

    torch.manual_seed(3)
    net = D.ContextUnet(3, 32, 4, bottleneck_k=4, dtype=torch.float32).to(DEV).train()
    g = torch.Generator().manual_seed(8)
    x = torch.rand(2, 3, 64, 64, generator=g).to(DEV)
    c = torch.randint(0, 4, (2,), generator=g).to(DEV)
    t = torch.rand(2, generator=g).to(DEV)
    m = torch.zeros(2).to(DEV)
    probe = torch.randn(2, 3, 64, 64, generator=g).to(DEV)

    def grads():
        return {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}

    for mod in net.modules():                       # BatchNorm buffers must not move between the two runs being compared
        if hasattr(mod, "momentum"):
            mod.momentum = 0.0
    net.zero_grad(set_to_none=True)
    (net(x, c, t, m) * probe).sum().backward()
    g1 = grads()
    net.zero_grad(set_to_none=True)
    out = (net(x, c, t, m) * probe).sum()
    try:
        out.backward(retain_graph=True)
        out.backward()
    except (DmError, RuntimeError) as e:            # loud refusal is an accepted outcome
        print("second backward refused:", str(e)[:120])
        return
    g2 = grads()
    assert g1.keys() == g2.keys()
    worst = max(((g2[k] - 2 * g1[k]).norm() / (2 * g1[k].norm() + 1e-20)).item() for k in g1 if g1[k].norm() > 1e-6)
    print("two backward passes: worst relative deviation from 2 x one pass", worst)
    assert worst <= 2e-3, worst


def test_lazy_zero_grad_overwriting_weight_gradients_match_the_full_fill():
    """r04: FusedAdamW.zero_grad leaves the gradient ranges of the layers on the halo weight-gradient kernel un-zeroed (marked fresh; their
    first launch overwrites: DmWgrad.overwrite -> the reduce launch writes instead of read-modify-writing) and zeroes the rest with one
    dm_zero_ranges launch.  Against the same model with the full fill (`_lazy_enabled = False`): per-step loss and pre-clip gradient
    norm over four steps incl. a two-micro-batch accumulation group; a fresh range nobody wrote is zeros after settle_fresh()."""
    import diffusionmodel_amd as D

    def make(lazy):
        torch.manual_seed(21)
        net = D.ContextUnet(3, 64, 4, bottleneck_k=4, dtype=torch.bfloat16)
        ddpm = D.DDPM(net, (1e-4, 0.02), 1, DEV, drop_prob=0.0)
        ddpm.train()
        ddpm.rng_seed = 5
        opt = D.FusedAdamW(ddpm.parameters(), lr=1e-3, weight_decay=1e-5, max_grad_norm=1.0)
        opt._lazy_enabled = lazy
        return ddpm, opt

    g = torch.Generator().manual_seed(3)
    x = torch.rand(4, 3, 64, 64, generator=g).to(DEV)
    c = torch.randint(0, 4, (4,), generator=g).to(DEV)
    am = torch.ones(4, 64, 64).to(DEV)

    def run(lazy):
        ddpm, opt = make(lazy)
        rec = []
        for step in range(4):
            opt.zero_grad()
            n_micro = 2 if step == 2 else 1                     # one accumulation group: the second micro-batch must ADD
            for m in range(n_micro):
                loss = ddpm(x[m::n_micro], c[m::n_micro], am[m::n_micro]) / n_micro
                loss.backward()
            opt.step()
            rec.append((loss.item(), opt.grad_norm().item()))
        return rec, ddpm, opt

    ra, da, oa = run(False)
    rb, db, ob = run(True)
    assert oa._lazy is None and ob._lazy is not None
    table, n_rows, lazy = ob._lazy
    lazy_elems = sum(p.numel() for p in lazy)
    assert len(lazy) >= 20 and lazy_elems > 0.5 * ob.total, (len(lazy), lazy_elems, ob.total)      # the 3x3 layers: most of the buffer
    for (la, na), (lb, nb) in zip(ra, rb):
        assert abs(la - lb) <= 5e-3 * max(abs(la), 1e-3), (ra, rb)          # (a stale or double-counted range moves these by tens of per cent)
        assert abs(na - nb) <= 5e-2 * na, (ra, rb)
    # a fresh range that no launch writes: garbage until settle_fresh(), zeros after (what an idle data-parallel rank contributes)
    ob.zero_grad()
    lazy[0].main_grad.fill_(7.0)
    assert lazy[0]._dm_fresh
    ob.settle_fresh()
    assert float(lazy[0].main_grad.abs().max()) == 0.0 and not lazy[0]._dm_fresh
