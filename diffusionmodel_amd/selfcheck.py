"""smoke(): one tiny train step (DDPM.forward + backward) and two CFG sampling steps of the HIP path on
cuda:0, checked against the CPU oracle (oracle/ is test infrastructure; this is one of the three places
allowed to import it)."""
import torch


def smoke(verbose=True):
    from oracle import synth, unet_ref as O
    from . import ContextUnet, DDPM
    dev = "cuda:0"
    nf, ncls, S, B, n_T = 32, 4, 64, 2, 1000
    spec = O.context_unet_spec(3, nf, ncls, 4)
    state = synth.synth_state(spec)
    net = ContextUnet(3, nf, ncls, bottleneck_k=4, dtype=torch.float32)
    net.load_state_dict(state)
    ddpm = DDPM(net, (1e-4, 0.02), n_T, dev, drop_prob=0.1)
    x = synth.synth_input("smoke.x", (B, 3, S, S))
    c = torch.tensor([1, 3])
    am = synth.synth_attn_mask(B, S)
    ts = torch.tensor([97, 723])
    keep = torch.tensor([1.0, 0.0])
    noise = synth.synth_noise("smoke.noise", (B, 3, S, S))
    ddpm.train()
    loss = ddpm(x.to(dev), c.to(dev), am.to(dev), ts=ts.to(dev), noise=noise.to(dev), ctx_mask=keep.to(dev))
    loss.backward()
    g_hip = net.out[3].weight.grad.detach().cpu()
    # oracle
    P = {"nn_model." + k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
         for k, v in state.items()}
    sched = O.ddpm_schedules(1e-4, 0.02, n_T)
    lo = O.ddpm_loss(P, sched, n_T, x, c, am, ts, noise, keep, True)
    lo.backward()
    g_ref = P["nn_model.out.3.weight"].grad
    err_l = abs(loss.item() - lo.item())
    err_g = (g_hip - g_ref).abs().max().item() / max(g_ref.abs().max().item(), 1e-12)
    if verbose:
        print(f"[smoke] train loss hip={loss.item():.6f} oracle={lo.item():.6f} |d|={err_l:.2e}; out.3 wgrad rel err {err_g:.2e}")
    assert err_l < 5e-4 and err_g < 5e-3, (err_l, err_g)
    # sampling: 2 steps of a 5-step schedule
    P2 = {"nn_model." + k: v.clone() for k, v in state.items()}
    net2 = ContextUnet(3, nf, ncls, bottleneck_k=4, dtype=torch.float32)
    net2.load_state_dict(state)
    d2 = DDPM(net2, (1e-4, 0.02), 5, dev, drop_prob=0.0)
    d2.eval()
    x_T = synth.synth_noise("smoke.z0", (4, 3, S, S))
    zs = [synth.synth_noise(f"smoke.z{j + 1}", (4, 3, S, S)) for j in range(5)]
    xs = d2.sample(4, (3, S, S), dev, guide_w=2.0, x_T=x_T, zs=zs, steps=2).cpu()
    with torch.no_grad():
        xr = O.ddpm_sample(P2, O.ddpm_schedules(1e-4, 0.02, 5), 5, ncls, x_T, zs, 2.0, steps=2)
    err_s = (xs - xr).abs().max().item()
    if verbose:
        print(f"[smoke] 2 sampling steps max |dx| = {err_s:.2e}")
    assert err_s < 1e-3, err_s
    return dict(loss_err=err_l, wgrad_rel_err=err_g, sample_err=err_s)
