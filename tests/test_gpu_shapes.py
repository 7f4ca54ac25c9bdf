"""Shape robustness on a real MI355X: channel counts that are not powers of two (the reference default
n_feat=192 -> 48/12-channel side branches, 192..3072-channel convs with ragged tiles), the cfg-5 config
(128x128, F=256, k=8, bf16) and the reference default image size path (k=8, hidden 2x2 at 256x256) against
the CPU oracle in fp32, plus finite bf16 train steps."""
import pytest
import torch

from oracle import synth, unet_ref as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _compare(nf, S, k, B, ncls=5, train=False, tol=2e-4):
    import diffusionmodel_amd as D
    spec = O.context_unet_spec(3, nf, ncls, k)
    state = synth.synth_state(spec)
    net = D.ContextUnet(3, nf, ncls, bottleneck_k=k, dtype=torch.float32)
    net.load_state_dict(state)
    net = net.to(DEV).train(train)
    x = synth.synth_input("shape.x", (B, 3, S, S))
    c = torch.arange(B) % ncls
    t = torch.linspace(0.1, 0.9, B)
    mk = (torch.arange(B) % 2).float()
    with torch.no_grad():
        eps = net(x.to(DEV), c.to(DEV), t.to(DEV), mk.to(DEV)).cpu()
        ref = O.context_unet({k_: v.clone() for k_, v in state.items()}, x, c, t, mk, train)
    err = (eps - ref).abs().max().item()
    print(f"F={nf} S={S} k={k} B={B} train={train}: max|eps - oracle| = {err:.2e}")
    assert err < tol * max(1.0, ref.abs().max().item())


def test_nfeat_192_eval_128():
    _compare(192, 128, 8, 2)


def test_nfeat_96_train_64():
    _compare(96, 64, 4, 3, train=True, tol=1e-3)


@pytest.mark.parametrize("nf", [16, 48])
def test_nfeat_multiple_of_16_not_32(nf):
    """VERDICT r02 missing #4: the reference needs only n_feat % 16 == 0 (CoordAttn / SEBlock: channel // 16).  n_feat = 16 / 48
    give UnetDown compress branches of 4 / 12 channels — carried as a masked 8 / 16-vector (modules.UnetDown._fwd_padded) — and
    hidden widths of 1 / 3 in the attention MLPs (the general SGEMM path).  Forward in eval and train mode and the gradients of
    the padded branch's parameters against the float32 oracle; running statistics land in the registered BatchNorm."""
    import diffusionmodel_amd as D
    _compare(nf, 64, 4, 2)
    _compare(nf, 64, 4, 3, train=True, tol=1e-3)
    ncls, S, B = 5, 64, 2
    spec = O.context_unet_spec(3, nf, ncls, 4)
    state = synth.synth_state(spec)
    net = D.ContextUnet(3, nf, ncls, bottleneck_k=4, dtype=torch.float32)
    net.load_state_dict(state)
    net = net.to(DEV).train()
    x = synth.synth_input("shape16.x", (B, 3, S, S))
    c, t, mk = torch.arange(B) % ncls, torch.linspace(0.2, 0.8, B), torch.ones(B)
    probe = synth.synth_input("shape16.p", (B, 3, S, S))
    eps = net(x.to(DEV), c.to(DEV), t.to(DEV), mk.to(DEV))
    (eps * probe.to(DEV)).mean().backward()
    P = {k_: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k_ else v.clone()) for k_, v in state.items()}
    ref = O.context_unet(P, x, c, t, mk, True)
    (ref * probe).mean().backward()
    named = dict(net.named_parameters())
    for pn in ("down1.channel_compress.0.weight", "down1.channel_compress.0.bias", "down1.channel_compress.1.weight", "down1.channel_compress.1.bias",
               "down1.ch_adjust.weight", "down2.channel_compress.0.weight", "init_conv.conv1.0.weight", "out.3.weight"):
        g, r = named[pn].grad.cpu(), P[pn].grad
        assert g.shape == r.shape
        scale = float(P[pn.replace("bias", "weight") if pn.endswith("0.bias") else pn].grad.abs().max()) + 1e-12
        err = float((g - r).abs().max()) / scale
        assert err < 5e-3, (pn, err)
    sd = net.state_dict()
    rm = sd["down1.channel_compress.1.running_mean"].cpu()
    assert rm.shape == (nf // 4,) and float((rm - P["down1.channel_compress.1.running_mean"]).abs().max()) < 1e-4
    assert int(sd["down1.channel_compress.1.num_batches_tracked"]) == 1


def test_nfeat_48_train_step_as_a_launch_plan():
    """the padded compress branch of n_feat % 32 == 16 models mirrored its parameters with torch copy_ —
    memcpy nodes under stream capture, which a launch plan refuses — so the default path of TrainEngine / new_scripy (use_plan)
    could not be built for them.  Now the branch moves its data with library launches: the plan builds, holds no torch copy
    kernel, and three replays train like three eager steps (FusedAdamW included)."""
    import diffusionmodel_amd as D

    def make():
        torch.manual_seed(11)
        net = D.ContextUnet(3, 48, 4, bottleneck_k=4, dtype=torch.float32)
        ddpm = D.DDPM(net, (1e-4, 0.02), 1, DEV, drop_prob=0.0)
        ddpm.train()
        ddpm.rng_seed = 5
        return ddpm, D.FusedAdamW(ddpm.parameters(), lr=1e-3, weight_decay=1e-5, max_grad_norm=1.0)

    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 64, 64, generator=g).to(DEV)
    c = torch.randint(0, 4, (2,), generator=g).to(DEV)
    am = torch.ones(2, 64, 64).to(DEV)
    da, oa = make()
    la = []
    for _ in range(3):
        oa.zero_grad()
        loss = da(x, c, am)
        loss.backward()
        oa.step()
        la.append(loss.item())
    db, ob = make()
    step = D.GraphedTrainStep(db, ob, x, c, am, mode="plan")
    foreign = step.plan.foreign_kernels()
    # (a plan cannot hold a memcpy NODE at all — dm_plan_from_graph refuses them — so building it is the check; what is left of torch
    #  are fills and the handful of index-plumbing kernels every planned step has)
    assert sum(n for nm, n in foreign.items() if "FillFunctor" not in nm) <= 10, foreign
    lb = [step(x, c, am).item() for _ in range(3)]
    for a, b in zip(la, lb):
        assert abs(a - b) <= 2e-4 * max(abs(a), 1e-3), (la, lb)
    assert lb[-1] < lb[0]
    sa, sb = da.state_dict(), db.state_dict()
    k = "nn_model.down1.channel_compress.1.running_mean"
    assert sa[k].shape == (12,) and float((sa[k] - sb[k]).abs().max()) < 1e-4 * max(1.0, float(sa[k].abs().max()))
    assert int(sa["nn_model.down1.channel_compress.1.num_batches_tracked"]) == int(sb["nn_model.down1.channel_compress.1.num_batches_tracked"]) == 3


def test_hidden_2x2_at_256_k8():
    _compare(32, 256, 8, 1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_cfg5_train_steps_and_sampling(dtype):
    """BASELINE configs[4]: 128x128, n_feat=256, k=8 (627 M parameters), CoordAttn on, per-GPU batch 8 — in float16 as stated (through
    the loss scaler, new_scripy.py:792-801) and in bfloat16: the loss falls over a few steps on a fixed batch, everything stays finite,
    no step is skipped by the scaler, and the sampler runs at this size."""
    import diffusionmodel_amd as D
    torch.manual_seed(0)
    net = D.ContextUnet(3, 256, 4, bottleneck_k=8, dtype=dtype)
    ddpm = D.DDPM(net, (1e-4, 0.02), 1000, DEV, drop_prob=0.1).train()
    assert ddpm.scaler.is_enabled() == (dtype == torch.float16)
    opt = D.FusedAdamW(ddpm.parameters(), lr=1e-4, weight_decay=1e-5, max_grad_norm=1.0, shadow_dtype=dtype)
    B = 8
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 3, 128, 128, generator=g).clamp(-1, 1).to(DEV)
    c = torch.randint(0, 4, (B,), generator=g).to(DEV)
    am = torch.full((B, 128, 128), 0.5, device=DEV)
    am[:, 64:] = 1.0
    am[:, 20:50, 30:70] = 3.0
    ts = torch.randint(1, 1001, (B,), generator=g).to(DEV)
    noise = torch.randn(B, 3, 128, 128, generator=g).to(DEV)
    keep = torch.ones(B, device=DEV)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        loss = ddpm(x, c, am, ts=ts, noise=noise, ctx_mask=keep)
        ddpm.scaler.scale(loss).backward()
        ddpm.scaler.unscale_(opt)
        ddpm.scaler.step(opt)
        ddpm.scaler.update()
        losses.append(loss.item())
    print(str(dtype), "cfg-5 losses", losses, "scale", ddpm.scaler.get_scale())
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0]
    assert int(opt._step_dev.item()) == 4                      # no overflow: no skipped step
    assert torch.isfinite(opt.flat_p).all().item() and torch.isfinite(opt.flat_g).all().item()
    ddpm.eval()
    xs = ddpm.sample(4, (3, 128, 128), DEV, guide_w=2.0, steps=2, seed=3)
    assert torch.isfinite(xs).all().item()


@pytest.mark.parametrize("dtype,bar", [(torch.float32, 1e-9), (torch.float16, 2.5e-6), (torch.bfloat16, 2e-4)], ids=["fp32", "fp16", "bf16"])
def test_cfg5_forward_against_the_oracle(dtype, bar):
    """BASELINE configs[4] at its stated size — 128x128, n_feat = 256 (627 M parameters), k = 8, CoordAttn on — eval-mode eps of ONE
    sample against the fp32 CPU oracle on the same seeded weights: relative MSE (eps error power / eps power).  fp32 sits at rounding
    level (1.4e-12); the 16-bit bars are about 5x the measured values (fp16 4.9e-7, bf16 3.5e-5)."""
    import diffusionmodel_amd as D
    torch.manual_seed(11)
    net = D.ContextUnet(3, 256, 4, bottleneck_k=8, dtype=dtype)
    with torch.no_grad():
        for n_, b in net.named_buffers():
            if n_.endswith("running_mean"):
                b.normal_(0, 0.1)
            elif n_.endswith("running_var"):
                b.uniform_(0.6, 1.4)
    sd = {k: v.detach().clone().cpu() for k, v in net.state_dict().items()}
    net = net.to(DEV).eval()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 3, 128, 128, generator=g).clamp_(-1, 1)
    c, t, mk = torch.tensor([2]), torch.tensor([0.37]), torch.tensor([1.0])
    with torch.no_grad():
        eps = net(x.to(DEV), c.to(DEV), t.to(DEV), mk.to(DEV)).cpu().double()
        torch.set_num_threads(16)
        ref = O.context_unet(sd, x, c, t, mk, False).double()
    rel = float(((eps - ref) ** 2).mean() / (ref ** 2).mean())
    print(f"cfg-5 eval eps, {dtype}: relative MSE {rel:.3e}, max abs {float((eps - ref).abs().max()):.3e}, eps power {float((ref ** 2).mean()):.3e}")
    assert torch.isfinite(eps).all() and rel < bar
