#!/usr/bin/env python3
"""Generate tests/golden/*.npz + schema.json by running the imported REFERENCE on closed-form data.

CONTAINER-ONLY: needs /root/reference (read-only mount) — it does not exist on the GPU box, where
the committed fixtures are used instead.  Run from the repo root:  python tests/golden/make_golden.py

What is pinned: schedule tables; per-module forward outputs + input/parameter
gradients in train() and eval(); whole ContextUnet (F=16, 4 classes) at 64^2 (k=4) and 128^2 (k=8);
DDPM.forward loss with the three random draws injected; DDPM.sample short trajectories with
injected noise; the MNIST ancestor net.  Weights/inputs come from oracle.synth (closed form), so
the fixtures hold outputs only, plus the key schema the reference's state_dict() exposes.

Harness adjustments that are not behaviour changes: `local_enhance` inside the net is
replaced by identity (the reference call is identically zero whenever it does not raise), and for
64^2 the bottleneck AvgPool/ConvTranspose kernel 8 is swapped for 4.
"""
import json
import os
import sys
from unittest import mock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import _refload, synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)
SCHEMA = {}


def load_synth(mod, prefix=""):
    sd = mod.state_dict()
    new = {k: synth.synth_tensor(prefix + k, tuple(v.shape)) for k, v in sd.items()}
    mod.load_state_dict(new, strict=True)
    return [(k, list(v.shape)) for k, v in sd.items()]


def run_module(name, mod, inputs, call=None, train_modes=(False, True)):
    """inputs: dict name->tensor (float ones get requires_grad).  Saves y, input grads, param grads,
    and BN running stats after the train-mode forward."""
    SCHEMA[name] = load_synth(mod)
    init = {k: v.clone() for k, v in mod.state_dict().items()}
    out = {}
    for train in train_modes:
        mod.load_state_dict(init)
        mod.train(train)
        tag = "train" if train else "eval"
        ins = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.startswith("mask") else v)
               for k, v in inputs.items()}
        y = call(mod, ins) if call else mod(*ins.values())
        probe = synth.synth_input(name + ".probe", tuple(y.shape))
        mod.zero_grad()
        (y * probe).sum().backward()
        out[f"{tag}.y"] = y.detach().numpy()
        for k, v in ins.items():
            if v.is_floating_point() and v.grad is not None:
                out[f"{tag}.d_{k}"] = v.grad.numpy()
        for k, p in mod.named_parameters():
            out[f"{tag}.g.{k}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        if train:
            for k, b in mod.named_buffers():
                if "running" in k:
                    out[f"train.buf.{k}"] = b.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in list(out.items())[:3]})


def gen_schedules(R):
    out = {}
    for T in (400, 700, 1000):
        for k, v in R.ddpm_schedules(1e-4, 0.02, T).items():
            out[f"T{T}.{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "schedules.npz"), **out)


def gen_modules(R):
    si = synth.synth_input
    run_module("se32", R.SEBlock(32), {"x": si("se32.x", (2, 32, 8, 8))})
    run_module("ca32_8", R.CoordAttn(32), {"x": si("ca32_8.x", (3, 32, 8, 8))})
    run_module("ca32_16", R.CoordAttn(32), {"x": si("ca32_16.x", (2, 32, 16, 16))})
    run_module("rcb_3_16_res", R.ResConvBlock(3, 16, True), {"x": si("rcb_3_16_res.x", (2, 3, 16, 16))})
    run_module("rcb_16_16_res", R.ResConvBlock(16, 16, True), {"x": si("rcb_16_16_res.x", (2, 16, 16, 16))})
    run_module("rcb_16_16_plain", R.ResConvBlock(16, 16, False), {"x": si("rcb_16_16_plain.x", (2, 16, 16, 16))})
    run_module("down_32_64", R.UnetDown(32, 64), {"x": si("down_32_64.x", (2, 32, 16, 16))})
    run_module("up_64_16", R.UnetUp(64, 16), {"x": si("up_64_16.x", (2, 32, 8, 8)), "skip": si("up_64_16.skip", (2, 32, 8, 8))})
    run_module("fc_1_32", R.EmbedFC(1, 32), {"x": si("fc_1_32.x", (5, 1))})
    run_module("fc_4_32", R.EmbedFC(4, 32), {"x": si("fc_4_32.x", (5, 4))})
    m = synth.synth_attn_mask(2, 16)
    run_module("le16", R.LocalEnhancer(16), {"x": si("le16.x", (2, 16, 16, 16)), "mask": m})


def make_ref_unet(R, nf, ncls, k):
    net = R.ContextUnet(3, nf, ncls)
    net.local_enhance.forward = lambda x, mask: x
    if k != 8:
        net.to_vec[0] = torch.nn.AvgPool2d(k)
        net.up0[0] = torch.nn.ConvTranspose2d(8 * nf, 8 * nf, k, k)
    return net


def child_grad_norms(net):
    d = {}
    for cname, child in net.named_children():
        sq = sum(float((p.grad.double() ** 2).sum()) for p in child.parameters() if p.grad is not None)
        d[cname] = sq ** 0.5
    return d


def gen_unet(R, tag, S, k, nf=32, ncls=4, B=2):
    net = make_ref_unet(R, nf, ncls, k)
    SCHEMA[tag] = load_synth(net)
    init = {kk: v.clone() for kk, v in net.state_dict().items()}
    x = synth.synth_input(tag + ".x", (B, 3, S, S))
    c = torch.tensor([(3 * i + 1) % ncls for i in range(B)])
    t = torch.tensor([(0.37 + 0.41 * i) % 1.0 for i in range(B)])
    mk = torch.tensor([float((i + 1) % 2) for i in range(B)])
    out = {"c": c.numpy(), "t": t.numpy(), "ctx_mask": mk.numpy()}
    for train in (False, True):
        net.load_state_dict(init)
        net.train(train)
        mode = "train" if train else "eval"
        xx = x.clone().requires_grad_(True)
        eps = net(xx, c, t, mk)
        probe = synth.synth_input(tag + ".probe", tuple(eps.shape))
        net.zero_grad()
        loss = (eps * probe).mean()
        loss.backward()
        out[f"{mode}.eps"] = eps.detach().numpy()
        out[f"{mode}.loss"] = np.float64(loss.item())
        out[f"{mode}.dx"] = xx.grad.numpy()
        for cn, v in child_grad_norms(net).items():
            out[f"{mode}.gn.{cn}"] = np.float64(v)
        for pn in ("out.3.weight", "init_conv.conv1.0.weight", "ca2.gamma_h", "ca1.alpha", "down1.ch_adjust.weight",
                   "time_emb2.model.0.weight", "up0.1.weight", "down2.down.3.se.fc.0.weight", "ctx_emb1.model.2.bias"):
            out[f"{mode}.g.{pn}"] = dict(net.named_parameters())[pn].grad.numpy()
        # the same reference code run in float64: separates rounding noise from real differences
        net.load_state_dict(init)
        net.double()
        with torch.no_grad():
            out[f"{mode}.eps64"] = net(x.double(), c, t.double(), mk.double()).numpy()
        net.float()
        net.load_state_dict(init)
        if train:
            net(x, c, t, mk)   # redo the fp32 train forward so the running stats below are the fp32 ones
            for bn in ("init_conv.conv1.1.running_mean", "init_conv.conv1.1.running_var",
                       "ca3.bn1_h.running_var", "up4.model.2.conv2.1.running_mean"):
                out[f"train.buf.{bn}"] = net.state_dict()[bn].numpy().copy()
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **out)
    print(tag, out["eval.eps"].shape, out["eval.loss"], out["train.loss"], "fp32-vs-fp64 noise eval/train:",
          np.abs(out["eval.eps"] - out["eval.eps64"]).max(), np.abs(out["train.eps"] - out["train.eps64"]).max())


class Inject:
    """Replace the reference's three RNG draws (new_scripy.py:405,406,413 / 445,465) by closed-form data."""

    def __init__(self, tag, n_T):
        self.tag, self.n_T, self.n = tag, n_T, 0

    def randint(self, lo, hi, shape, **kw):
        return torch.tensor([(lo + (313 * i + 96) % (hi - lo)) for i in range(shape[0])])

    def randn_like(self, x, **kw):
        return synth.synth_noise(self.tag + ".noise", tuple(x.shape))

    def bernoulli(self, p, **kw):
        return torch.tensor([float((i % 3) != 1) for i in range(p.shape[0])])

    def randn(self, *shape, **kw):
        self.n += 1
        return synth.synth_noise(f"{self.tag}.z{self.n - 1}", tuple(shape))


def gen_ddpm_forward(R, tag="ddpm_fwd64", S=64, k=4, nf=32, ncls=4, B=4, n_T=1000):
    net = make_ref_unet(R, nf, ncls, k)
    ddpm = R.DDPM(net, (1e-4, 0.02), n_T, "cpu", drop_prob=0.1)
    SCHEMA[tag] = load_synth(ddpm)   # includes the 7 schedule buffers -> overwritten by synth! restore:
    for kk, v in R.ddpm_schedules(1e-4, 0.02, n_T).items():
        getattr(ddpm, kk).copy_(v)
    x = synth.synth_input(tag + ".x", (B, 3, S, S))
    c = torch.tensor([(i + 2) % ncls for i in range(B)])
    am = synth.synth_attn_mask(B, S)
    inj = Inject(tag, n_T)
    out = {}
    init = {kk: v.clone() for kk, v in ddpm.state_dict().items()}
    for train in (True, False):
        ddpm.load_state_dict(init)
        ddpm.train(train)
        ddpm.zero_grad()
        with mock.patch.object(torch, "randint", inj.randint), mock.patch.object(torch, "randn_like", inj.randn_like), \
                mock.patch.object(torch, "bernoulli", inj.bernoulli):
            loss = ddpm(x, c, am)
        mode = "train" if train else "eval"
        out[f"{mode}.loss"] = np.float64(loss.item())
        if train:
            loss.backward()
            for cn, v in child_grad_norms(ddpm.nn_model).items():
                out[f"train.gn.{cn}"] = np.float64(v)
            out["train.g.out.3.weight"] = ddpm.nn_model.out[3].weight.grad.numpy()
    out["ts"] = inj.randint(1, n_T + 1, (B,)).numpy()
    out["keep"] = inj.bernoulli(torch.ones(B)).numpy()
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **out)
    print(tag, out['train.loss'], out['eval.loss'], out['ts'])


def gen_ddpm_sample(R, tag, S, k, n_T, n, guide_w, nf=32, ncls=4):
    net = make_ref_unet(R, nf, ncls, k)
    ddpm = R.DDPM(net, (1e-4, 0.02), n_T, "cpu", drop_prob=0.0)
    load_synth(ddpm)
    for kk, v in R.ddpm_schedules(1e-4, 0.02, n_T).items():
        getattr(ddpm, kk).copy_(v)
    ddpm.eval()
    inj = Inject(tag, n_T)
    with torch.no_grad(), mock.patch.object(torch, "randn", inj.randn):
        x = ddpm.sample(n, (3, S, S), "cpu", guide_w=guide_w)
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), x=x.numpy(), n_draws=np.int64(inj.n))
    print(tag, x.shape, float(x.abs().max()), inj.n)


def gen_mnist(M):
    nf, ncls = 16, 10
    net = M.ContextUnet(1, nf, ncls)
    SCHEMA["mnist16"] = load_synth(net)
    init = {k: v.clone() for k, v in net.state_dict().items()}
    B = 3
    x = synth.synth_input("mnist16.x", (B, 1, 28, 28))
    c = torch.tensor([1, 7, 4])
    t = torch.tensor([0.2, 0.55, 0.9])
    mk = torch.tensor([0.0, 1.0, 0.0])
    out = {}
    for train in (False, True):
        net.load_state_dict(init)
        net.train(train)
        mode = "train" if train else "eval"
        eps = net(x, c, t, mk)
        probe = synth.synth_input("mnist16.probe", tuple(eps.shape))
        net.zero_grad()
        (eps * probe).mean().backward()
        out[f"{mode}.eps"] = eps.detach().numpy()
        for cn, v in child_grad_norms(net).items():
            out[f"{mode}.gn.{cn}"] = np.float64(v)
    # DDPM.forward (plain MSE, MNIST_script.py:234-252) with injected draws
    net.load_state_dict(init)
    ddpm = M.DDPM(net, (1e-4, 0.02), 400, "cpu", drop_prob=0.1)
    inj = Inject("mnist16", 400)
    ddpm.train()
    with mock.patch.object(torch, "randint", inj.randint), mock.patch.object(torch, "randn_like", inj.randn_like), \
            mock.patch.object(torch, "bernoulli", lambda p, **kw: torch.tensor([float(i % 3 == 1) for i in range(p.shape[0])])):
        loss = ddpm(x, c)
    out["ddpm.loss"] = np.float64(loss.item())
    # sample: 4 steps, n=10 (labels are hard-coded arange(0,10), MNIST_script.py:262)
    net.load_state_dict(init)
    d2 = M.DDPM(net, (1e-4, 0.02), 4, "cpu", drop_prob=0.1)
    d2.eval()
    inj = Inject("mnist16.s", 4)
    with torch.no_grad(), mock.patch.object(torch, "randn", inj.randn):
        xs, store = d2.sample(10, (1, 28, 28), "cpu", guide_w=0.5)
    out["sample.x"] = xs.numpy()
    out["sample.store_shape"] = np.array(store.shape)
    np.savez_compressed(os.path.join(OUT, "mnist16.npz"), **out)
    print("mnist16", out["eval.eps"].shape, out["ddpm.loss"])


def main():
    assert _refload.available(), "reference not mounted; fixtures are generated in the authoring container only"
    R = _refload.load("new_scripy")
    M = _refload.load("MNIST_script")
    gen_schedules(R)
    gen_modules(R)
    gen_unet(R, "unet32_64", 64, 4)
    gen_unet(R, "unet32_128", 128, 8)
    gen_ddpm_forward(R)
    gen_ddpm_sample(R, "sample64_T5", 64, 4, 5, 4, 2.0)
    gen_ddpm_sample(R, "sample64_T3_w0", 64, 4, 3, 8, 0.0)
    gen_mnist(M)
    gen_metrics_and_masks(R)
    gen_bf16_autocast(R)
    gen_bf16_autocast(R, lp=torch.float16, fname="fp16_autocast.npz")
    gen_train3(R)
    gen_train3(R, autocast=True)
    gen_f128_b2(R)
    # key schema of the full-size nets (shapes only; no tensors are instantiated for the big ones)
    SCHEMA["ddpm_keys_F32_k4"] = [(k, list(v.shape)) for k, v in
                                  R.DDPM(make_ref_unet(R, 32, 4, 4), (1e-4, 0.02), 1000, "cpu").state_dict().items()]
    SCHEMA["unet_keys_F32_k8_c10"] = [(k, list(v.shape)) for k, v in R.ContextUnet(3, 32, 10).state_dict().items()]
    SCHEMA["mnist_keys_F32"] = [(k, list(v.shape)) for k, v in M.ContextUnet(1, 32, 10).state_dict().items()]
    SCHEMA["cfg"] = {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(R.Cfg).items() if k.isupper()}
    with open(os.path.join(OUT, "schema.json"), "w") as f:
        json.dump(SCHEMA, f, indent=0)
    sz = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("total fixture bytes", sz)




def gen_metrics_and_masks(R):
    """ImageMetrics.calc_ssim / calc_psnr / evaluate_batch (static parts) and CrackDataset's mask rasterisation,
    the latter by running the real dataset class on a throw-away folder with two tiny PNGs + VOC XML files."""
    import tempfile
    from PIL import Image
    out = {}
    pairs = []
    for i in range(6):
        a = synth.synth_input(f"metrics.a{i}", (3, 32, 32))
        b = a + 0.2 * synth.synth_input(f"metrics.b{i}", (3, 32, 32))
        if i % 2 == 1:                      # already in [0, 1]: no rescale branch
            a, b = (a.clamp(-1, 1) + 1) / 2, (b.clamp(-1, 1) + 1) / 2
        if i == 4:
            b = (b.clamp(-1, 1) + 1) / 2    # mixed ranges: only img1 is rescaled
        pairs.append((a, b))
    out["ssim"] = np.array([R.ImageMetrics.calc_ssim(a, b) for a, b in pairs], dtype=np.float64)
    out["psnr"] = np.array([R.ImageMetrics.calc_psnr(a, b) for a, b in pairs], dtype=np.float64)
    im = R.ImageMetrics.__new__(R.ImageMetrics)       # __init__ wants the pretrained InceptionV3 (remote fetch)
    real = torch.stack([p[0] for p in pairs[:4]])
    gen = torch.stack([p[1] for p in pairs[:4]])
    ev = im.evaluate_batch(real, gen)
    out["eval.ssim"], out["eval.psnr"] = np.float64(ev["ssim"]), np.float64(ev["psnr"])
    # masks
    boxes = [(13, 7, 57, 40, 100, 80), (0, 0, 300, 200, 300, 200), (5, 150, 25, 199, 64, 200), (33, 21, 34, 22, 67, 45)]
    old = R.Cfg.IMG_SIZE
    try:
        for S in (64, 256):
            R.Cfg.IMG_SIZE = S
            with tempfile.TemporaryDirectory() as d:
                os.makedirs(os.path.join(d, "images", "crack"))
                os.makedirs(os.path.join(d, "annotations"))
                for j, (x0, y0, x1, y1, w, h) in enumerate(boxes):
                    Image.new("RGB", (w, h)).save(os.path.join(d, "images", "crack", f"im{j}.png"))
                    with open(os.path.join(d, "annotations", f"im{j}.xml"), "w") as f:
                        f.write(f"<annotation><size><width>{w}</width><height>{h}</height></size><object><bndbox>"
                                f"<xmin>{x0}</xmin><ymin>{y0}</ymin><xmax>{x1}</xmax><ymax>{y1}</ymax></bndbox></object></annotation>")
                ds = R.CrackDataset(d, transform=None)
                order = {os.path.basename(s[0]): k for k, s in enumerate(ds.samples)}
                for j in range(len(boxes)):
                    _, label, mask = ds[order[f"im{j}.png"]]
                    out[f"mask.S{S}.{j}"] = mask.numpy()
    finally:
        R.Cfg.IMG_SIZE = old
    out["boxes"] = np.array(boxes)
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **out)
    print("metrics", out["ssim"], out["psnr"])


def _per_child_grads(net):
    """{child: flat float64 gradient vector} over the top-level children of a net (zeros where a parameter got no gradient)."""
    d = {}
    for cname, child in net.named_children():
        vs = [(p.grad if p.grad is not None else torch.zeros_like(p)).detach().double().reshape(-1) for p in child.parameters()]
        if vs:
            d[cname] = torch.cat(vs)
    return d


def _cos(a, b):
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))


def gen_bf16_autocast(R, S=64, k=4, nf=32, ncls=4, lp=torch.bfloat16, fname="bf16_autocast.npz"):
    """The REFERENCE under torch.autocast("cpu", dtype=torch.bfloat16) — the precision mode BASELINE configs[1] is quoted in
    (new_scripy.py:784 wraps DDPM.forward in autocast) — next to the same reference code in float64, on the unet32_64 and
    ddpm_fwd64 cases.  Called a second time with lp=torch.float16 -> fp16_autocast.npz (the dtype torch.cuda.amp.autocast() picks on
    a GPU, i.e. what the reference actually trains in there; the key names keep the "_bf16" suffix = "the autocast run").
    Pins what "bf16 parity" means: the reference's own bf16 error (eps MSE, loss, per-child gradient
    norm ratio and cosine against fp64) is the yardstick the HIP bf16 path is held to (tests/test_gpu_bf16.py)."""
    out = {}
    # ---- (a) denoiser alone, eval + train
    tag, B = "unet32_64", 2
    net = make_ref_unet(R, nf, ncls, k)
    load_synth(net)
    init = {kk: v.clone() for kk, v in net.state_dict().items()}
    x = synth.synth_input(tag + ".x", (B, 3, S, S))
    c = torch.tensor([(3 * i + 1) % ncls for i in range(B)])
    t = torch.tensor([(0.37 + 0.41 * i) % 1.0 for i in range(B)])
    mk = torch.tensor([float((i + 1) % 2) for i in range(B)])
    probe = synth.synth_input(tag + ".probe", (B, 3, S, S))
    for train in (False, True):
        mode = "train" if train else "eval"
        # float64 run of the reference: eps, loss, per-child gradients
        net.load_state_dict(init)
        net.double()
        net.train(train)
        net.zero_grad()
        e64 = net(x.double(), c, t.double(), mk.double())
        l64 = (e64 * probe.double()).mean()
        l64.backward()
        g64 = _per_child_grads(net)
        net.float()
        net.load_state_dict(init)
        net.train(train)
        net.zero_grad()
        with torch.autocast("cpu", dtype=lp):
            e16 = net(x, c, t, mk)
            l16 = (e16.float() * probe).mean()
        # float16: backward under the GradScaler's initial loss scale (new_scripy.py:390,792: scaler.scale(loss).backward(), then
        # unscale_) — unscaled fp16 gradients of this size underflow; bfloat16 needs no scaling
        ls = 65536.0 if lp == torch.float16 else 1.0
        (l16 * ls).backward()
        g16 = {cn: v / ls for cn, v in _per_child_grads(net).items()}
        e16 = e16.detach().float()
        out[f"unet.{mode}.eps_bf16"] = e16.numpy()
        out[f"unet.{mode}.eps64"] = e64.detach().numpy()
        out[f"unet.{mode}.mse_bf16_vs_64"] = np.float64(((e16.double() - e64.detach()) ** 2).mean().item())
        out[f"unet.{mode}.maxabs_bf16_vs_64"] = np.float64((e16.double() - e64.detach()).abs().max().item())
        out[f"unet.{mode}.power64"] = np.float64((e64.detach() ** 2).mean().item())
        out[f"unet.{mode}.loss64"] = np.float64(l64.item())
        out[f"unet.{mode}.loss_bf16"] = np.float64(l16.item())
        for cn in g64:
            out[f"unet.{mode}.gn64.{cn}"] = np.float64(g64[cn].norm().item())
            out[f"unet.{mode}.gn_bf16.{cn}"] = np.float64(g16[cn].norm().item())
            out[f"unet.{mode}.cos_bf16.{cn}"] = np.float64(_cos(g16[cn], g64[cn]))
        print("bf16 autocast", mode, "mse", out[f"unet.{mode}.mse_bf16_vs_64"], "max", out[f"unet.{mode}.maxabs_bf16_vs_64"],
              "loss", l64.item(), l16.item(), "min cos", min(out[f"unet.{mode}.cos_bf16.{cn}"] for cn in g64 if cn != "local_enhance"))
    # ---- (b) DDPM.forward with the injected draws (same case as ddpm_fwd64.npz)
    tag, B, n_T = "ddpm_fwd64", 4, 1000
    net = make_ref_unet(R, nf, ncls, k)
    ddpm = R.DDPM(net, (1e-4, 0.02), n_T, "cpu", drop_prob=0.1)
    load_synth(ddpm)
    for kk, v in R.ddpm_schedules(1e-4, 0.02, n_T).items():
        getattr(ddpm, kk).copy_(v)
    init = {kk: v.clone() for kk, v in ddpm.state_dict().items()}
    x = synth.synth_input(tag + ".x", (B, 3, S, S))
    c = torch.tensor([(i + 2) % ncls for i in range(B)])
    am = synth.synth_attn_mask(B, S)
    inj = Inject(tag, n_T)
    for train in (True, False):
        mode = "train" if train else "eval"
        res = {}
        for prec in ("64", "bf16"):
            ddpm.float()
            ddpm.load_state_dict(init)
            net.__dict__.pop("forward", None)
            if prec == "64":
                ddpm.double()
                # harness only: DDPM.forward hands the net a float32 t and keep-mask (new_scripy.py:413-415); the float64 run casts them
                net.forward = lambda xx, cc, tt, mm, _f=type(net).forward, _n=net: _f(_n, xx, cc, tt.double(), mm.double())
            ddpm.train(train)
            ddpm.zero_grad()
            xin = x.double() if prec == "64" else x
            with mock.patch.object(torch, "randint", inj.randint), \
                    mock.patch.object(torch, "randn_like", lambda v, **kw: inj.randn_like(v).to(v.dtype)), \
                    mock.patch.object(torch, "bernoulli", lambda p, **kw: inj.bernoulli(p).to(p.dtype)):
                if prec == "bf16":
                    with torch.autocast("cpu", dtype=lp):                  # new_scripy.py:784
                        loss = ddpm(xin, c, am)
                else:
                    loss = ddpm(xin, c, am.double())
            res[prec] = float(loss.item())
            if train:
                ls = 65536.0 if (prec == "bf16" and lp == torch.float16) else 1.0
                (loss * ls).backward()
                res["g" + prec] = {cn: v / ls for cn, v in _per_child_grads(ddpm.nn_model).items()}
        ddpm.float()
        net.__dict__.pop("forward", None)
        out[f"ddpm.{mode}.loss64"], out[f"ddpm.{mode}.loss_bf16"] = np.float64(res["64"]), np.float64(res["bf16"])
        if train:
            for cn in res["g64"]:
                out[f"ddpm.train.gn64.{cn}"] = np.float64(res["g64"][cn].norm().item())
                out[f"ddpm.train.gn_bf16.{cn}"] = np.float64(res["gbf16"][cn].norm().item())
                out[f"ddpm.train.cos_bf16.{cn}"] = np.float64(_cos(res["gbf16"][cn], res["g64"][cn]))
        print("bf16 autocast ddpm", mode, res["64"], res["bf16"])
    np.savez_compressed(os.path.join(OUT, fname), **out)


def gen_train3(R, S=64, k=4, nf=32, ncls=4, B=2, n_T=1000, accum=2, n_opt=3, lr=1e-4, wd=1e-5, autocast=False):
    """Three optimiser steps of the reference's train loop (new_scripy.py:777-803: loss / ACCUM_STEPS, backward, every ACCUM_STEPS
    micro-batches clip_grad_norm_(1.0) + AdamW.step + zero_grad) on the CPU in fp32, with every random draw injected
    (micro-batch m uses the tags train3.m<m>.*).  Pins the training driver + fused optimiser: per-micro-batch losses, the
    pre-clip gradient norm of every step, per-child parameter norms after the last step, torch's AdamW state of three
    parameters, and two parameter tensors in full."""
    net = make_ref_unet(R, nf, ncls, k)
    ddpm = R.DDPM(net, (1e-4, 0.02), n_T, "cpu", drop_prob=0.1)
    load_synth(ddpm)
    for kk, v in R.ddpm_schedules(1e-4, 0.02, n_T).items():
        getattr(ddpm, kk).copy_(v)
    ddpm.train()
    optim = torch.optim.AdamW(ddpm.parameters(), lr=lr, weight_decay=wd)
    old = R.Cfg.ACCUM_STEPS
    R.Cfg.ACCUM_STEPS = accum
    out = {"losses": [], "grad_norms": []}
    try:
        optim.zero_grad()
        for m in range(accum * n_opt):
            tag = f"train3.m{m}"
            x = synth.synth_input(tag + ".x", (B, 3, S, S))
            c = torch.tensor([(m + i) % ncls for i in range(B)])
            am = synth.synth_attn_mask(B, S)
            inj = Inject(tag, n_T)
            ts = torch.tensor([1 + (313 * (i + 2 * m) + 96) % n_T for i in range(B)])
            keep = torch.tensor([float(((i + m) % 3) != 1) for i in range(B)])
            with mock.patch.object(torch, "randint", lambda *a, **kw: ts), mock.patch.object(torch, "randn_like", inj.randn_like), \
                    mock.patch.object(torch, "bernoulli", lambda p, **kw: keep), \
                    torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):             # :784 (bf16 variant: train3_bf16.npz)
                loss = ddpm(x, c, am) / R.Cfg.ACCUM_STEPS                                   # :785-786
            out["losses"].append(loss.item() * R.Cfg.ACCUM_STEPS)                          # :789
            loss.backward()
            if (m + 1) % R.Cfg.ACCUM_STEPS == 0:                                           # :795-803
                out["grad_norms"].append(float(torch.nn.utils.clip_grad_norm_(ddpm.parameters(), 1.0)))
                optim.step()
                optim.zero_grad()
            out[f"ts.{m}"], out[f"keep.{m}"] = ts.numpy(), keep.numpy()
    finally:
        R.Cfg.ACCUM_STEPS = old
    res = {"losses": np.array(out["losses"]), "grad_norms": np.array(out["grad_norms"]),
           "hyper": np.array([lr, wd, accum, n_opt, B])}
    if autocast:       # the reference's own bf16 trajectory: only what the tolerance of the HIP bf16 run is derived from
        np.savez_compressed(os.path.join(OUT, "train3_bf16.npz"), **res)
        print("train3 (autocast bf16) losses", res["losses"], "grad norms", res["grad_norms"])
        return
    for m in range(accum * n_opt):
        res[f"ts.{m}"], res[f"keep.{m}"] = out[f"ts.{m}"], out[f"keep.{m}"]
    for cn, child in ddpm.nn_model.named_children():
        ps = [p.detach().double().reshape(-1) for p in child.parameters()]
        if ps:
            res[f"pnorm.{cn}"] = np.float64(torch.cat(ps).norm().item())
    named = dict(ddpm.named_parameters())
    for pn in ("nn_model.out.3.weight", "nn_model.down2.down.0.weight", "nn_model.ca2.gamma_h", "nn_model.time_emb1.model.0.bias"):
        res[f"p.{pn}"] = named[pn].detach().numpy()
        st = optim.state[named[pn]]
        res[f"m.{pn}"], res[f"v.{pn}"] = st["exp_avg"].numpy(), st["exp_avg_sq"].numpy()
        res[f"step.{pn}"] = np.float64(float(st["step"]))
    sd = ddpm.state_dict()
    for bn in ("nn_model.init_conv.conv1.1.running_mean", "nn_model.up4.model.2.conv2.1.running_var"):
        res[f"buf.{bn}"] = sd[bn].numpy().copy()
    res["nbt"] = np.int64(int(sd["nn_model.init_conv.conv1.1.num_batches_tracked"]))
    # torch's optimizer.state_dict() layout, for the FusedAdamW.state_dict() schema test
    osd = optim.state_dict()
    SCHEMA["adamw_state_dict"] = {"param_group_keys": sorted(osd["param_groups"][0].keys()), "n_params": len(osd["param_groups"][0]["params"]),
                                  "state_keys": sorted(osd["state"][0].keys()),
                                  "state0": {kk: (list(v.shape), str(v.dtype)) for kk, v in osd["state"][0].items()}}
    np.savez_compressed(os.path.join(OUT, "train3.npz"), **res)
    print("train3 losses", res["losses"], "grad norms", res["grad_norms"])


def gen_f128_b2(R, S=64, k=4, nf=128, ncls=4, B=2):
    """The REFERENCE at the benchmark's width (BASELINE configs[1]: n_feat = 128, 64x64, k = 4), B = 2 -> f128_b2.npz: eps in
    float64, float32 and under torch.autocast("cpu", bfloat16), eval and train; the probe loss; per-child gradient norms (fp64) and
    the autocast run's norm / cosine against fp64.  Pins the HIP path AT THE BENCHMARK WIDTH to the reference itself (ADVICE r02:
    until now F = 128 was held to the oracle only)."""
    tag = "f128_b2"
    net = make_ref_unet(R, nf, ncls, k)
    load_synth(net)
    init = {kk: v.clone() for kk, v in net.state_dict().items()}
    x = synth.synth_input(tag + ".x", (B, 3, S, S))
    c = torch.tensor([(3 * i + 1) % ncls for i in range(B)])
    t = torch.tensor([(0.37 + 0.41 * i) % 1.0 for i in range(B)])
    mk = torch.tensor([float((i + 1) % 2) for i in range(B)])
    probe = synth.synth_input(tag + ".probe", (B, 3, S, S))
    out = {"c": c.numpy(), "t": t.numpy(), "ctx_mask": mk.numpy()}
    for train in (False, True):
        mode = "train" if train else "eval"
        net.load_state_dict(init)
        net.double()
        net.train(train)
        net.zero_grad()
        e64 = net(x.double(), c, t.double(), mk.double())
        l64 = (e64 * probe.double()).mean()
        l64.backward()
        g64 = _per_child_grads(net)
        net.float()
        net.load_state_dict(init)
        net.train(train)
        net.zero_grad()
        with torch.no_grad():
            e32 = net(x, c, t, mk)
        net.load_state_dict(init)
        net.train(train)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            e16 = net(x, c, t, mk)
            l16 = (e16.float() * probe).mean()
        l16.backward()
        g16 = _per_child_grads(net)
        e16 = e16.detach().float()
        out[f"{mode}.eps64"] = e64.detach().numpy()
        out[f"{mode}.eps32_maxabs_vs_64"] = np.float64((e32.double() - e64.detach()).abs().max().item())
        out[f"{mode}.mse_bf16_vs_64"] = np.float64(((e16.double() - e64.detach()) ** 2).mean().item())
        out[f"{mode}.maxabs_bf16_vs_64"] = np.float64((e16.double() - e64.detach()).abs().max().item())
        out[f"{mode}.power64"] = np.float64((e64.detach() ** 2).mean().item())
        out[f"{mode}.loss64"] = np.float64(l64.item())
        out[f"{mode}.loss_bf16"] = np.float64(l16.item())
        for cn in g64:
            out[f"{mode}.gn64.{cn}"] = np.float64(g64[cn].norm().item())
            out[f"{mode}.gn_bf16.{cn}"] = np.float64(g16[cn].norm().item())
            out[f"{mode}.cos_bf16.{cn}"] = np.float64(_cos(g16[cn], g64[cn]))
        print("f128_b2", mode, "fp32 max|e32-e64|", out[f"{mode}.eps32_maxabs_vs_64"], "autocast mse", out[f"{mode}.mse_bf16_vs_64"],
              "power", out[f"{mode}.power64"], "loss", l64.item(), l16.item())
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **out)


def gen_f128_b2_band(R, S=64, k=4, nf=128, ncls=4, B=2, K=5):
    """How far the REFERENCE's own autocast(bfloat16) run moves per-child gradient norms at the benchmark width is one rounding-noise
    realisation per input: the f128_b2 case again on K inputs that differ from it by 1e-3 of noise (x + 1e-3 * synth_noise(
    "f128_b2.pert<j>")) -> f128_b2_band.npz with, per input j and mode, the per-child gradient norms of the float64 run and of
    the autocast run and the eps MSE of the autocast run.  (r04: on the unperturbed input the autocast run's ca3 error is 0.030,
    on these five 0.004 .. 0.14 — a bar of '1.25 x the single realisation' for the HIP path's own single realisation is a coin
    toss; tests/test_gpu_bf16.py compares the two error DISTRIBUTIONS over the K + 1 inputs instead.)"""
    tag = "f128_b2"
    net = make_ref_unet(R, nf, ncls, k)
    load_synth(net)
    init = {kk: v.clone() for kk, v in net.state_dict().items()}
    x0 = synth.synth_input(tag + ".x", (B, 3, S, S))
    c = torch.tensor([(3 * i + 1) % ncls for i in range(B)])
    t = torch.tensor([(0.37 + 0.41 * i) % 1.0 for i in range(B)])
    mk = torch.tensor([float((i + 1) % 2) for i in range(B)])
    probe = synth.synth_input(tag + ".probe", (B, 3, S, S))
    out = {"K": np.int64(K), "pert_scale": np.float64(1e-3)}
    for j in range(1, K + 1):
        x = x0 + 1e-3 * synth.synth_noise(f"{tag}.pert{j}", (B, 3, S, S))
        for train in (False, True):
            mode = "train" if train else "eval"
            net.load_state_dict(init)
            net.double()
            net.train(train)
            net.zero_grad()
            e64 = net(x.double(), c, t.double(), mk.double())
            l64 = (e64 * probe.double()).mean()
            l64.backward()
            g64 = _per_child_grads(net)
            net.float()
            net.load_state_dict(init)
            net.train(train)
            net.zero_grad()
            with torch.autocast("cpu", dtype=torch.bfloat16):
                e16 = net(x, c, t, mk)
                l16 = (e16.float() * probe).mean()
            l16.backward()
            g16 = _per_child_grads(net)
            e16 = e16.detach().float()
            out[f"{j}.{mode}.mse_bf16_vs_64"] = np.float64(((e16.double() - e64.detach()) ** 2).mean().item())
            out[f"{j}.{mode}.loss64"] = np.float64(l64.item())
            out[f"{j}.{mode}.eps64_sum"] = np.float64(e64.detach().sum().item())
            for cn in g64:
                out[f"{j}.{mode}.gn64.{cn}"] = np.float64(g64[cn].norm().item())
                out[f"{j}.{mode}.gn_bf16.{cn}"] = np.float64(g16[cn].norm().item())
            errs = {cn: g16[cn].norm().item() / g64[cn].norm().item() - 1 for cn in g64 if g64[cn].norm().item() > 0}
            print("f128_b2_band", j, mode, "autocast mse", out[f"{j}.{mode}.mse_bf16_vs_64"], "worst child", max(errs, key=lambda q: abs(errs[q])),
                  round(max(abs(v) for v in errs.values()), 4), flush=True)
    np.savez_compressed(os.path.join(OUT, tag + "_band.npz"), **out)


if __name__ == "__main__":
    if os.environ.get("DM_GOLDEN_ONLY") == "r04":          # the round-4 addition only
        gen_f128_b2_band(_refload.load("new_scripy"))
    elif os.environ.get("DM_GOLDEN_ONLY") == "r03":          # the round-3 addition only
        gen_f128_b2(_refload.load("new_scripy"))
    elif os.environ.get("DM_GOLDEN_ONLY") == "metrics":
        gen_metrics_and_masks(_refload.load("new_scripy"))
    elif os.environ.get("DM_GOLDEN_ONLY") == "r02":        # the round-2 additions only (schema.json gets the new key merged in)
        R_ = _refload.load("new_scripy")
        gen_bf16_autocast(R_)
        gen_bf16_autocast(R_, lp=torch.float16, fname="fp16_autocast.npz")
        gen_train3(R_)
        gen_train3(R_, autocast=True)
        with open(os.path.join(OUT, "schema.json")) as f:
            full = json.load(f)
        full.update(SCHEMA)
        with open(os.path.join(OUT, "schema.json"), "w") as f:
            json.dump(full, f, indent=0)
    else:
        main()
