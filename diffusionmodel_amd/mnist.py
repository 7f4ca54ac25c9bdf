"""The MNIST ancestor (reference: MNIST_script.py:31-300) on the same HIP operator layer: 2-level
ContextUnet with MaxPool downs and ConvTranspose 2x2 ups, plain-MSE DDPM, standard (flipped-mask) CFG.
Constructor spellings follow MNIST_script.py (`in_channels=`); state_dict keys match its schema.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import ACT_RELU, DmError
from .ddpm import DDPM as _DDPM
from .ddpm import repeat2
from .modules import EmbedFC, ResConvBlock, _conv, _HipBlock, _pad8, _ToNHWC


class ResidualConvBlock(ResConvBlock):
    """MNIST_script.py:31-65 — no SE block."""

    def __init__(self, in_channels, out_channels, is_res=False):
        super().__init__(in_channels, out_channels, is_res, with_se=False)
        self.same_channels = self.same_ch


class UnetDown(_HipBlock):
    """MNIST_script.py:68-78: ResidualConvBlock + MaxPool2d(2)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.out_ch = out_channels
        self.model = nn.Sequential(ResidualConvBlock(in_channels, out_channels), nn.MaxPool2d(2))

    def _out_channels(self):
        return self.out_ch

    def _fwd(self, x):
        return ops.MaxPool2.apply(self.model[0]._fwd(x))

    def forward(self, x):
        return self._nchw_call(x)


class UnetUp(_HipBlock):
    """MNIST_script.py:81-97: cat -> ConvTranspose2d(2,2) -> 2 blocks."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.out_ch = out_channels
        self.model = nn.Sequential(nn.ConvTranspose2d(in_channels, out_channels, 2, 2),
                                   ResidualConvBlock(out_channels, out_channels), ResidualConvBlock(out_channels, out_channels))
        self.model[0].weight.data = self.model[0].weight.data.contiguous(memory_format=torch.channels_last)

    def _out_channels(self):
        return self.out_ch

    def _fwd(self, x, skip):
        u = ops.Cat.apply(x, skip)
        u = ops.ConvTransposeKS.apply(u, self.model[0].weight, self.model[0].bias, self.compute_dtype)
        return self.model[2]._fwd(self.model[1]._fwd(u))

    def forward(self, x, skip):
        xs = _ToNHWC.apply(x, self.compute_dtype, x.shape[1])
        ss = _ToNHWC.apply(skip, self.compute_dtype, skip.shape[1])
        from .modules import _ToNCHW
        return _ToNCHW.apply(self._fwd(xs, ss), self.out_ch)


class ContextUnet(_HipBlock):
    """MNIST_script.py:119-187."""

    def __init__(self, in_channels, n_feat=256, n_classes=10, bottleneck_k=7, dtype=None):
        super().__init__()
        if n_feat % 8:
            raise DmError("n_feat must be a multiple of 8 on the HIP path")
        self.in_channels, self.n_feat, self.n_classes, self.bottleneck_k = in_channels, n_feat, n_classes, bottleneck_k
        F = n_feat
        self.init_conv = ResidualConvBlock(in_channels, F, is_res=True)
        self.down1, self.down2 = UnetDown(F, F), UnetDown(F, 2 * F)
        self.to_vec = nn.Sequential(nn.AvgPool2d(bottleneck_k), nn.GELU())
        self.timeembed1, self.timeembed2 = EmbedFC(1, 2 * F), EmbedFC(1, F)
        self.contextembed1, self.contextembed2 = EmbedFC(n_classes, 2 * F), EmbedFC(n_classes, F)
        self.up0 = nn.Sequential(nn.ConvTranspose2d(2 * F, 2 * F, bottleneck_k, bottleneck_k), nn.GroupNorm(8, 2 * F), nn.ReLU())
        self.up0[0].weight.data = self.up0[0].weight.data.contiguous(memory_format=torch.channels_last)
        self.up1, self.up2 = UnetUp(4 * F, F), UnetUp(2 * F, F)
        self.out = nn.Sequential(_conv(2 * F, F, 3, 1, 1), nn.GroupNorm(8, F), nn.ReLU(), _conv(F, in_channels, 3, 1, 1))
        self._sp_out0 = ops.ConvSpec(3, 3, 1, 1)
        self._sp_out3 = ops.ConvSpec(3, 3, 1, 1, out_nchw=True)
        self.set_compute_dtype(torch.float32 if dtype is None else dtype)

    def _out_channels(self):
        return self.in_channels

    def encode(self, x):
        ops.L.require_device(x)
        return self._encode(_ToNHWC.apply(x, self.compute_dtype, _pad8(self.in_channels)))

    def _encode(self, x8):
        x0 = self.init_conv._fwd(x8)
        d1 = self.down1._fwd(x0)
        d2 = self.down2._fwd(d1)
        hidden = ops.AvgPoolGelu.apply(d2, self.bottleneck_k)
        u1 = ops.ConvTransposeKS.apply(hidden, self.up0[0].weight, self.up0[0].bias, self.compute_dtype)
        u1 = ops.GroupNormAct.apply(u1, self.up0[1].weight, self.up0[1].bias, 8, ACT_RELU)
        return x0, d1, d2, u1

    def embed(self, c, t, context_mask):
        oh = ops.onehot_mask(c.long(), context_mask.float(), self.n_classes, flip=True)      # MNIST_script.py:168-171
        t = t.reshape(-1, 1).float()
        return self.contextembed1(oh), self.timeembed1(t), self.contextembed2(oh), self.timeembed2(t)

    def decode(self, feats, embs):
        x0, d1, d2, u1 = feats
        cemb1, temb1, cemb2, temb2 = embs
        u2 = self.up1._fwd(ops.Film.apply(u1, cemb1, temb1), d2)
        u3 = self.up2._fwd(ops.Film.apply(u2, cemb2, temb2), d1)
        y = ops.conv_bn_act(u3, x0, self.out[0], None, self._sp_out0)
        y = ops.GroupNormAct.apply(y, self.out[1].weight, self.out[1].bias, 8, ACT_RELU)
        return ops.conv_bn_act(y, None, self.out[3], None, self._sp_out3)

    def forward(self, x, c, t, context_mask):
        return self.decode(self.encode(x), self.embed(c, t, context_mask))


class DDPM(_DDPM):
    """MNIST_script.py:219-300: plain MSE loss, drop mask ~ Bernoulli(drop_prob) with 1 = drop, sample()
    returns (x, x_i_store)."""

    def forward(self, x, c, *, ts=None, noise=None, context_mask=None):
        ops.L.require_device(x)
        dev = x.device
        B = x.shape[0]
        tfrac = None
        if ts is None or context_mask is None or noise is None:
            if getattr(self, "_rng_dev", None) is None or self._rng_dev.device != dev:
                self._rng_dev = torch.full((1,), self._rng_calls, dtype=torch.int64, device=dev)
            self._rng_calls += 1
            ts_d = torch.empty(B, dtype=torch.int64, device=dev)
            tf_d, drop_d = torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.float32, device=dev)
            # MNIST_script.py:237, 246: t ~ U{1..n_T}; context_mask ~ Bernoulli(drop_prob) with 1 = drop
            ops.call("dm_draw_ts_keep", ops.ptr(ts_d), ops.ptr(tf_d), ops.ptr(drop_d), B, int(self.n_T), float(self.drop_prob),
                     self._seed(), ops.ptr(self._rng_dev))
            if ts is None:
                ts, tfrac = ts_d, tf_d
            if context_mask is None:
                context_mask = drop_d
            if noise is None:
                noise = ops.randn(tuple(x.shape), dev, self._seed(), self._rng_dev)
        ts = ts.to(dev).long()
        if tfrac is None:
            tfrac = ts.float() / self.n_T
        net = self.nn_model
        xt = ops.qsample(x.float(), noise, ts, self.sqrtab, self.sqrtmab, net.compute_dtype, _pad8(x.shape[1]))
        pred = net.decode(net._encode(xt), net.embed(c.to(dev), tfrac, context_mask.to(dev)))
        return ops.WeightedLoss.apply(pred, noise, None, None)

    def _eps_cfg(self, x_i, c2, mask2, t2, ctx_embs, dedup):
        net = self.nn_model
        temb1, temb2 = net.timeembed1(t2.reshape(-1, 1)), net.timeembed2(t2.reshape(-1, 1))
        embs = (ctx_embs[0], temb1, ctx_embs[1], temb2)
        if dedup and not net.training:
            feats = tuple(repeat2(f) for f in net.encode(x_i))
        else:
            feats = net.encode(repeat2(x_i))
        return net.decode(feats, embs)

    @torch.no_grad()
    def sample(self, n_sample, size, device, guide_w=0.0, *, x_T=None, zs=None, dedup=True, seed=None, steps=None):
        net = self.nn_model
        dev = torch.device(device)
        seed = self._sample_seed() if seed is None else int(seed)       # unseeded calls draw fresh noise, like the reference
        x_i = (x_T.to(dev).float().contiguous().clone() if x_T is not None else ops.randn((n_sample,) + tuple(size), dev, seed, 0))
        c_i = torch.arange(0, self.n_classes, device=dev).repeat(n_sample // self.n_classes).repeat(2)
        mask = torch.zeros(2 * n_sample, device=dev)
        mask[n_sample:] = 1.0                                            # second half context-free (MNIST_script.py:271)
        oh = ops.onehot_mask(c_i, mask, self.n_classes, flip=True)
        ctx_embs = (net.contextembed1(oh), net.contextembed2(oh))
        step = torch.full((1,), self.n_T, dtype=torch.int32, device=dev)
        t2 = torch.empty(2 * n_sample, dtype=torch.float32, device=dev)
        sched = {k: getattr(self, k) for k in ("oneover_sqrta", "mab_over_sqrtmab", "sqrt_beta_t")}
        store = []
        n_iter = self.n_T if steps is None else steps
        for j in range(n_iter):
            i = self.n_T - j
            ops.fill_t(t2, step, self.n_T)
            eps = self._eps_cfg(x_i, c_i, mask, t2, ctx_embs, dedup)
            z = zs[j].to(dev).float().contiguous() if (zs is not None and i > 1) else None
            ops.cfg_update(x_i, eps, z, guide_w, sched, step, seed=seed, dec_step=True)
            if i % 20 == 0 or i == self.n_T or i < 8:                    # MNIST_script.py:296-297
                store.append(x_i.detach().cpu().numpy())
        return x_i, np.array(store)
