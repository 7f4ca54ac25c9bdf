"""Functional CPU restatement of the reference denoiser and DDPM wrapper.  TEST INFRASTRUCTURE.

Everything here is a pure function of a flat parameter dictionary `P` whose keys and shapes are the
reference's `state_dict()` schema (so reference checkpoints, the HIP modules and this oracle all
speak the same names).  The math follows /root/reference/new_scripy.py and MNIST_script.py; each
function cites the lines it restates.  Pinned against the imported reference by tests/golden/
(see tests/golden/make_golden.py and tests/test_oracle_golden.py).

Conventions: activations NCHW, dtype = dtype of `P` (fp32 by default, fp64 for noise-floor
studies), BatchNorm in train mode updates `P[...running_*]` in place exactly like nn.BatchNorm2d.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d / nn.GroupNorm defaults
BN_MOM = 0.1


# ----------------------------------------------------------------------------- leaf helpers
def _conv(x, P, pre, stride=1, padding=0):
    return F.conv2d(x, P[pre + ".weight"], P.get(pre + ".bias"), stride=stride, padding=padding)


def _bn(x, P, pre, train):
    """nn.BatchNorm2d semantics (biased var to normalise, unbiased into running_var)."""
    if train and (pre + ".num_batches_tracked") in P:
        P[pre + ".num_batches_tracked"] += 1
    return F.batch_norm(x, P[pre + ".running_mean"], P[pre + ".running_var"],
                        P[pre + ".weight"], P[pre + ".bias"], train, BN_MOM, BN_EPS)


def _gn(x, P, pre, groups=8):
    return F.group_norm(x, groups, P[pre + ".weight"], P[pre + ".bias"], BN_EPS)


def _gelu(x):
    return F.gelu(x)  # exact erf form, nn.GELU() default


def _linear(x, P, pre):
    return F.linear(x, P[pre + ".weight"], P.get(pre + ".bias"))


# ----------------------------------------------------------------------------- blocks
def se_block(x, P, pre):
    """new_scripy.py:143-158 — squeeze (mean HW) -> fc(no bias) -> GELU -> fc(no bias) -> sigmoid -> scale."""
    b, c = x.shape[:2]
    y = x.mean(dim=(2, 3))
    y = torch.sigmoid(_linear(_gelu(_linear(y, P, pre + ".fc.0")), P, pre + ".fc.2"))
    return x * y.view(b, c, 1, 1)


def coord_attn(x, P, pre, train):
    """new_scripy.py:97-140."""
    n, c, h, w = x.shape
    x_h = x.mean(dim=3, keepdim=True)                      # (n,c,h,1)   :102
    x_w = x.mean(dim=2, keepdim=True)                      # (n,c,1,w)   :103
    x_h = _gelu(_bn(_conv(x_h, P, pre + ".conv1_h"), P, pre + ".bn1_h", train))   # :105-107
    x_w = _gelu(_bn(_conv(x_w, P, pre + ".conv1_w"), P, pre + ".bn1_w", train))   # :109-111
    h2w = _conv(x_h, P, pre + ".h2w_proj").permute(0, 1, 3, 2)   # (n,c',1,h)  :113,116
    w2h = _conv(x_w, P, pre + ".w2h_proj").permute(0, 1, 3, 2)   # (n,c',w,1)  :114,117
    h2w = F.adaptive_avg_pool2d(h2w, (1, w))               # identity when h == w  :119
    w2h = F.adaptive_avg_pool2d(w2h, (h, 1))               # :120
    x_h = x_h + torch.sigmoid(P[pre + ".gamma_h"]) * w2h   # :122,125
    x_w = x_w + torch.sigmoid(P[pre + ".gamma_w"]) * h2w   # :123,126
    a_h = torch.sigmoid(_conv(x_h, P, pre + ".conv_h"))    # :128
    a_w = torch.sigmoid(_conv(x_w, P, pre + ".conv_w"))    # :129
    al = torch.sigmoid(P[pre + ".alpha"])
    be = torch.sigmoid(P[pre + ".beta"])
    s = al + be + 1e-8                                     # :134
    return x * ((al / s) * a_h + (be / s) * a_w)           # :138-140


def local_enhancer(x, mask, P, pre, high_thresh=1.2):
    """new_scripy.py:161-174, standalone contract: mask is (B,H,W)."""
    hi = (mask > high_thresh).to(x.dtype).unsqueeze(1)
    y = _conv(x, P, pre + ".conv.0", padding=1)
    y = _gelu(_gn(y, P, pre + ".conv.1"))
    y = _conv(y, P, pre + ".conv.3", padding=1)
    return x + y * hi


def _conv_bn_gelu(x, P, conv_pre, bn_pre, train):
    return _gelu(_bn(_conv(x, P, conv_pre, padding=1), P, bn_pre, train))


def res_conv_block(x, P, pre, is_res, train, with_se=True):
    """new_scripy.py:176-209 (with_se) / MNIST_script.py:31-65 (no SE).  Note the literal 1.414."""
    x1 = _conv_bn_gelu(x, P, pre + ".conv1.0", pre + ".conv1.1", train)
    x2 = _conv_bn_gelu(x1, P, pre + ".conv2.0", pre + ".conv2.1", train)
    if not is_res:
        return x2
    if with_se:
        x2 = se_block(x2, P, pre + ".se")
    same = P[pre + ".conv1.0.weight"].shape[0] == P[pre + ".conv1.0.weight"].shape[1]
    return ((x if same else x1) + x2) / 1.414


def unet_down(x, P, pre, train):
    """new_scripy.py:211-235."""
    x = _gelu(_bn(_conv(x, P, pre + ".channel_compress.0"), P, pre + ".channel_compress.1", train))
    x = _conv(x, P, pre + ".ch_adjust")
    x = _conv_bn_gelu(x, P, pre + ".down.0", pre + ".down.1", train)
    x = res_conv_block(x, P, pre + ".down.3", True, train)
    return _conv(x, P, pre + ".down.4", stride=2, padding=1)


def unet_up(x, skip, P, pre, train):
    """new_scripy.py:237-253."""
    x = torch.cat((x, skip), 1)
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    x = _conv(x, P, pre + ".model.0.1", padding=1)
    x = res_conv_block(x, P, pre + ".model.1", False, train)
    return res_conv_block(x, P, pre + ".model.2", False, train)


def embed_fc(x, P, pre):
    """new_scripy.py:255-268."""
    in_dim = P[pre + ".model.0.weight"].shape[1]
    x = x.reshape(-1, in_dim)
    return _linear(_gelu(_linear(x, P, pre + ".model.0")), P, pre + ".model.2")


def context_unet(P, x, c, t, ctx_mask, train=False, pre=""):
    """new_scripy.py:317-356.  The bottleneck kernel k is read off `up0.0.weight` (8 in the
    reference, 4 for 64x64 — SURVEY §0.4).  `local_enhance` contributes nothing inside the net
    (the reference's call is identically zero whenever it does not raise)."""
    nf = P[pre + "init_conv.conv1.0.weight"].shape[0]
    ncls = P[pre + "ctx_emb1.model.0.weight"].shape[1]
    k = P[pre + "up0.0.weight"].shape[2]
    dt = P[pre + "init_conv.conv1.0.weight"].dtype
    x0 = res_conv_block(x.to(dt), P, pre + "init_conv", True, train)
    d1 = coord_attn(unet_down(x0, P, pre + "down1", train), P, pre + "ca1", train)
    d2 = coord_attn(unet_down(d1, P, pre + "down2", train), P, pre + "ca2", train)
    d3 = coord_attn(unet_down(d2, P, pre + "down3", train), P, pre + "ca3", train)
    d4 = coord_attn(unet_down(d3, P, pre + "down4", train), P, pre + "ca4", train)
    hidden = _gelu(F.avg_pool2d(d4, k))                                        # :332
    onehot = F.one_hot(c.long(), ncls).to(dt) * ctx_mask.to(dt)[:, None]       # :334-340
    t = t.to(dt)
    cemb1 = embed_fc(onehot, P, pre + "ctx_emb1").view(-1, nf * 8, 1, 1)
    temb1 = embed_fc(t, P, pre + "time_emb1").view(-1, nf * 8, 1, 1)
    cemb2 = embed_fc(onehot, P, pre + "ctx_emb2").view(-1, nf * 4, 1, 1)
    temb2 = embed_fc(t, P, pre + "time_emb2").view(-1, nf * 4, 1, 1)
    u1 = F.conv_transpose2d(hidden, P[pre + "up0.0.weight"], P[pre + "up0.0.bias"], stride=k)
    u1 = F.relu(_gn(u1, P, pre + "up0.1"))                                     # :297-301
    u2 = unet_up(cemb1 * u1 + temb1, d4, P, pre + "up1", train)                # :348
    u3 = unet_up(cemb2 * u2 + temb2, d3, P, pre + "up2", train)                # :349
    u4 = unet_up(u3, d2, P, pre + "up3", train)
    u5 = unet_up(u4, d1, P, pre + "up4", train)
    y = _conv(torch.cat((u5, x0), 1), P, pre + "out.0", padding=1)             # :355
    y = F.relu(_gn(y, P, pre + "out.1"))
    return _conv(y, P, pre + "out.3", padding=1)


# ----------------------------------------------------------------------------- MNIST ancestor
def mnist_context_unet(P, x, c, t, context_mask, train=False, pre=""):
    """MNIST_script.py:155-187 — 2-level net, MaxPool downs, ConvT 2x2 ups, flipped mask (:170)."""
    nf = P[pre + "init_conv.conv1.0.weight"].shape[0]
    ncls = P[pre + "contextembed1.model.0.weight"].shape[1]
    k = P[pre + "up0.0.weight"].shape[2]
    dt = P[pre + "init_conv.conv1.0.weight"].dtype
    x0 = res_conv_block(x.to(dt), P, pre + "init_conv", True, train, with_se=False)
    d1 = F.max_pool2d(res_conv_block(x0, P, pre + "down1.model.0", False, train), 2)
    d2 = F.max_pool2d(res_conv_block(d1, P, pre + "down2.model.0", False, train), 2)
    hidden = _gelu(F.avg_pool2d(d2, k))
    m = -1 * (1 - context_mask.to(dt))[:, None]                                # :168-170
    onehot = F.one_hot(c.long(), ncls).to(dt) * m
    t = t.to(dt)
    cemb1 = embed_fc(onehot, P, pre + "contextembed1").view(-1, nf * 2, 1, 1)
    temb1 = embed_fc(t, P, pre + "timeembed1").view(-1, nf * 2, 1, 1)
    cemb2 = embed_fc(onehot, P, pre + "contextembed2").view(-1, nf, 1, 1)
    temb2 = embed_fc(t, P, pre + "timeembed2").view(-1, nf, 1, 1)
    u1 = F.conv_transpose2d(hidden, P[pre + "up0.0.weight"], P[pre + "up0.0.bias"], stride=k)
    u1 = F.relu(_gn(u1, P, pre + "up0.1"))

    def up(z, skip, p):
        z = torch.cat((z, skip), 1)
        z = F.conv_transpose2d(z, P[p + ".model.0.weight"], P[p + ".model.0.bias"], stride=2)
        z = res_conv_block(z, P, p + ".model.1", False, train)
        return res_conv_block(z, P, p + ".model.2", False, train)

    u2 = up(cemb1 * u1 + temb1, d2, pre + "up1")
    u3 = up(cemb2 * u2 + temb2, d1, pre + "up2")
    y = _conv(torch.cat((u3, x0), 1), P, pre + "out.0", padding=1)
    y = F.relu(_gn(y, P, pre + "out.1"))
    return _conv(y, P, pre + "out.3", padding=1)


# ----------------------------------------------------------------------------- DDPM wrapper
SCHEDULE_KEYS = ("alpha_t", "oneover_sqrta", "sqrt_beta_t", "alphabar_t", "sqrtab", "sqrtmab",
                 "mab_over_sqrtmab")


def ddpm_schedules(beta1, beta2, T):
    """new_scripy.py:358-384 (= MNIST_script.py:190-216).  fp32, op order log -> cumsum -> exp."""
    assert beta1 < beta2 < 1.0
    beta = (beta2 - beta1) * torch.arange(0, T + 1, dtype=torch.float32) / T + beta1
    alpha = 1 - beta
    abar = torch.cumsum(torch.log(alpha), dim=0).exp()
    sqrtmab = torch.sqrt(1 - abar)
    return OrderedDict(alpha_t=alpha, oneover_sqrta=1 / torch.sqrt(alpha),
                       sqrt_beta_t=torch.sqrt(beta), alphabar_t=abar, sqrtab=torch.sqrt(abar),
                       sqrtmab=sqrtmab, mab_over_sqrtmab=(1 - alpha) / sqrtmab)


class LossCfg:
    """Loss constants the reference reads from Cfg at call time (new_scripy.py:30-36, 420-435)."""
    HIGH_THRESH, MID_THRESH = 1.2, 0.8
    HIGH_WEIGHT, MID_WEIGHT, LOW_WEIGHT = 3.0, 1.0, 0.5
    FEAT_CONSIST_WEIGHT = 2.0


def q_sample(sched, x, ts, noise):
    """new_scripy.py:408-411."""
    return sched["sqrtab"][ts, None, None, None] * x + sched["sqrtmab"][ts, None, None, None] * noise


def weighted_loss(pred, noise, attn_mask, cfg=LossCfg):
    """new_scripy.py:417-437.  attn_mask (B,H,W) -> repeated to 3 channels (:418)."""
    m = attn_mask.to(pred.dtype).unsqueeze(1).expand(-1, pred.shape[1], -1, -1)
    w = torch.where(m > cfg.HIGH_THRESH, torch.tensor(cfg.HIGH_WEIGHT, dtype=pred.dtype),
                    torch.where(m > cfg.MID_THRESH, torch.tensor(cfg.MID_WEIGHT, dtype=pred.dtype),
                                torch.tensor(cfg.LOW_WEIGHT, dtype=pred.dtype)))
    hi = (m > cfg.HIGH_THRESH).to(pred.dtype)
    return (((noise - pred) ** 2) * w).mean() + (pred * hi - noise * hi).abs().mean() * cfg.FEAT_CONSIST_WEIGHT


def ddpm_loss(P, sched, n_T, x, c, attn_mask, ts, noise, keep_mask, train=True, pre="nn_model.",
              cfg=LossCfg):
    """DDPM.forward with the three random draws injected (new_scripy.py:401-439)."""
    x_t = q_sample(sched, x, ts, noise)
    pred = context_unet(P, x_t, c, ts / n_T, keep_mask, train, pre)
    return weighted_loss(pred, noise.to(pred.dtype), attn_mask, cfg)


def mnist_ddpm_loss(P, sched, n_T, x, c, ts, noise, drop_mask, train=True, pre="nn_model."):
    """MNIST_script.py:234-252 — plain MSE; drop_mask ~ Bernoulli(drop_prob), 1 = drop."""
    x_t = q_sample(sched, x, ts, noise)
    pred = mnist_context_unet(P, x_t, c, ts / n_T, drop_mask, train, pre)
    return F.mse_loss(pred, noise.to(pred.dtype))


def cfg_update(sched, i, x, eps1, eps2, guide_w, z):
    """new_scripy.py:468-475: eps = (1+w) eps[:n] - w eps[n:]; ancestral update."""
    eps = (1 + guide_w) * eps1 - guide_w * eps2
    return sched["oneover_sqrta"][i] * (x - eps * sched["mab_over_sqrtmab"][i]) + sched["sqrt_beta_t"][i] * z


def ddpm_sample(P, sched, n_T, n_classes, x_T, zs, guide_w, steps=None, pre="nn_model.",
                net=context_unet, trajectory=False):
    """DDPM.sample (new_scripy.py:441-477 / MNIST_script.py:254-300) with injected noise.

    x_T: (n,C,H,W) start; zs: sequence of per-step noise, zs[j] used at step i = n_T - j (ignored
    at i == 1).  Labels cycle 0..n_classes-1; first half of the doubled batch has ctx_mask 0,
    second half 1 (:450-454) — with new_scripy's net that makes eps1 the UNconditional branch
    (SURVEY §0.5), with the MNIST net (mask flipped at MNIST_script.py:170) the conditional one.
    `steps` limits the number of iterations (trajectory tests)."""
    n = x_T.shape[0]
    c = torch.arange(n_classes).repeat(n // n_classes).repeat(2)
    mask = torch.zeros(2 * n)
    mask[n:] = 1.0
    x = x_T
    traj = []
    last = 1 if steps is None else n_T - steps + 1
    for j, i in enumerate(range(n_T, last - 1, -1)):
        t = torch.full((2 * n,), i / n_T, dtype=torch.float32)
        eps = net(P, x.repeat(2, 1, 1, 1), c, t, mask, False, pre)
        z = zs[j] if i > 1 else torch.zeros_like(x)
        x = cfg_update(sched, i, x, eps[:n], eps[n:], guide_w, z).to(x_T.dtype)
        if trajectory:
            traj.append(x.clone())
    return (x, traj) if trajectory else x


# ----------------------------------------------------------------------------- key schema
def _rcb_spec(s, pre, cin, cout, se):
    for i, (a, b) in enumerate(((cin, cout), (cout, cout)), 1):
        s[f"{pre}.conv{i}.0.weight"] = (b, a, 3, 3)
        s[f"{pre}.conv{i}.0.bias"] = (b,)
        _bn_spec(s, f"{pre}.conv{i}.1", b)
    if se:
        s[f"{pre}.se.fc.0.weight"] = (cout // 16, cout)
        s[f"{pre}.se.fc.2.weight"] = (cout, cout // 16)


def _bn_spec(s, pre, c):
    s[pre + ".weight"] = (c,)
    s[pre + ".bias"] = (c,)
    s[pre + ".running_mean"] = (c,)
    s[pre + ".running_var"] = (c,)
    s[pre + ".num_batches_tracked"] = ()


def _conv_spec(s, pre, cout, cin, k):
    s[pre + ".weight"] = (cout, cin, k, k)
    s[pre + ".bias"] = (cout,)


def _fc_spec(s, pre, i, o):
    s[pre + ".model.0.weight"] = (o, i)
    s[pre + ".model.0.bias"] = (o,)
    s[pre + ".model.2.weight"] = (o, o)
    s[pre + ".model.2.bias"] = (o,)


def context_unet_spec(in_ch=3, n_feat=192, n_classes=10, k=8):
    """Ordered name -> shape map equal to reference ContextUnet(...).state_dict() (new_scripy.py:270-315)."""
    F_ = n_feat
    s = OrderedDict()
    _rcb_spec(s, "init_conv", in_ch, F_, True)
    for name, ci, co in (("down1", F_, F_), ("down2", F_, 2 * F_), ("down3", 2 * F_, 4 * F_), ("down4", 4 * F_, 8 * F_)):
        _conv_spec(s, name + ".channel_compress.0", ci // 4, ci, 1)
        _bn_spec(s, name + ".channel_compress.1", ci // 4)
        _conv_spec(s, name + ".ch_adjust", co, ci // 4, 1)
        _conv_spec(s, name + ".down.0", co, co, 3)
        _bn_spec(s, name + ".down.1", co)
        _rcb_spec(s, name + ".down.3", co, co, True)
        _conv_spec(s, name + ".down.4", co, co, 4)
    for name, ch in (("ca1", F_), ("ca2", 2 * F_), ("ca3", 4 * F_), ("ca4", 8 * F_)):
        r = ch // 16
        s[name + ".gamma_h"] = (1,)
        s[name + ".gamma_w"] = (1,)
        s[name + ".alpha"] = (1,)
        s[name + ".beta"] = (1,)
        _conv_spec(s, name + ".conv1_h", r, ch, 1)
        _conv_spec(s, name + ".conv1_w", r, ch, 1)
        _bn_spec(s, name + ".bn1_h", r)
        _bn_spec(s, name + ".bn1_w", r)
        _conv_spec(s, name + ".h2w_proj", r, r, 1)
        _conv_spec(s, name + ".w2h_proj", r, r, 1)
        _conv_spec(s, name + ".conv_h", ch, r, 1)
        _conv_spec(s, name + ".conv_w", ch, r, 1)
    _fc_spec(s, "time_emb1", 1, 8 * F_)
    _fc_spec(s, "time_emb2", 1, 4 * F_)
    _fc_spec(s, "ctx_emb1", n_classes, 8 * F_)
    _fc_spec(s, "ctx_emb2", n_classes, 4 * F_)
    s["up0.0.weight"] = (8 * F_, 8 * F_, k, k)
    s["up0.0.bias"] = (8 * F_,)
    s["up0.1.weight"] = (8 * F_,)
    s["up0.1.bias"] = (8 * F_,)
    for name, ci, co in (("up1", 16 * F_, 4 * F_), ("up2", 8 * F_, 2 * F_), ("up3", 4 * F_, F_), ("up4", 2 * F_, F_)):
        _conv_spec(s, name + ".model.0.1", co, ci, 3)
        _rcb_spec(s, name + ".model.1", co, co, False)
        _rcb_spec(s, name + ".model.2", co, co, False)
    _conv_spec(s, "local_enhance.conv.0", F_, F_, 3)
    s["local_enhance.conv.1.weight"] = (F_,)
    s["local_enhance.conv.1.bias"] = (F_,)
    _conv_spec(s, "local_enhance.conv.3", F_, F_, 3)
    _conv_spec(s, "out.0", F_, 2 * F_, 3)
    s["out.1.weight"] = (F_,)
    s["out.1.bias"] = (F_,)
    _conv_spec(s, "out.3", in_ch, F_, 3)
    return s


def mnist_unet_spec(in_channels=1, n_feat=256, n_classes=10, k=7):
    """Ordered name -> shape map equal to MNIST_script.ContextUnet(...).state_dict() (MNIST_script.py:119-153)."""
    F_ = n_feat
    s = OrderedDict()
    _rcb_spec(s, "init_conv", in_channels, F_, False)
    _rcb_spec(s, "down1.model.0", F_, F_, False)
    _rcb_spec(s, "down2.model.0", F_, 2 * F_, False)
    _fc_spec(s, "timeembed1", 1, 2 * F_)
    _fc_spec(s, "timeembed2", 1, F_)
    _fc_spec(s, "contextembed1", n_classes, 2 * F_)
    _fc_spec(s, "contextembed2", n_classes, F_)
    s["up0.0.weight"] = (2 * F_, 2 * F_, k, k)
    s["up0.0.bias"] = (2 * F_,)
    s["up0.1.weight"] = (2 * F_,)
    s["up0.1.bias"] = (2 * F_,)
    for name, ci, co in (("up1", 4 * F_, F_), ("up2", 2 * F_, F_)):
        s[name + ".model.0.weight"] = (ci, co, 2, 2)
        s[name + ".model.0.bias"] = (co,)
        _rcb_spec(s, name + ".model.1", co, co, False)
        _rcb_spec(s, name + ".model.2", co, co, False)
    _conv_spec(s, "out.0", F_, 2 * F_, 3)
    s["out.1.weight"] = (F_,)
    s["out.1.bias"] = (F_,)
    _conv_spec(s, "out.3", in_channels, F_, 3)
    return s
