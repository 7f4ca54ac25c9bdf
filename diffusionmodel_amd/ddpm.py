"""DDPM wrapper: noise schedules, training loss and classifier-free-guidance sampling
(reference: new_scripy.py:358-477) on the HIP operator layer.

`DDPM.forward(x, c, attn_mask)` and `DDPM.sample(n_sample, size, device, guide_w, refine_steps)` keep the
reference's signatures; keyword-only extras let tests inject the random draws (the reference never
seeds anything, so parity harnesses must) and switch on the MI355X-specific execution options:
encoder de-duplication under CFG (the down path does not depend on the class context)
and hipGraph replay of the per-step kernel sequence with a device-resident step counter.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops
from ._lib import DmError
from .config import Cfg
from .modules import _HipBlock, _pad8

SCHEDULE_KEYS = ("alpha_t", "oneover_sqrta", "sqrt_beta_t", "alphabar_t", "sqrtab", "sqrtmab", "mab_over_sqrtmab")


def ddpm_schedules(beta1, beta2, T):
    """The 7 fp32 tables of length T+1 (new_scripy.py:358-384).  Host code; the op order
    (log -> cumsum -> exp) is kept so the tables are bit-identical to the reference's."""
    if not (beta1 < beta2 < 1.0):
        raise AssertionError("beta1 and beta2 must be in (0, 1)")
    steps = torch.arange(0, T + 1, dtype=torch.float32)
    beta_t = (beta2 - beta1) * steps / T + beta1
    alpha_t = 1 - beta_t
    alphabar_t = torch.cumsum(torch.log(alpha_t), dim=0).exp()
    sqrtmab = torch.sqrt(1 - alphabar_t)
    tables = OrderedDict()
    tables["alpha_t"] = alpha_t
    tables["oneover_sqrta"] = 1 / torch.sqrt(alpha_t)
    tables["sqrt_beta_t"] = torch.sqrt(beta_t)
    tables["alphabar_t"] = alphabar_t
    tables["sqrtab"] = torch.sqrt(alphabar_t)
    tables["sqrtmab"] = sqrtmab
    tables["mab_over_sqrtmab"] = (1 - alpha_t) / sqrtmab
    return tables


def repeat2(x):
    """Batch-double a contiguous tensor with two device copies (CFG doubling)."""
    out = torch.empty((2 * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    n = x.numel()
    d = ops.dt(x)
    ops.call("dm_cast", ops.ptr(x), out.data_ptr(), d, d, n)
    ops.call("dm_cast", ops.ptr(x), out.data_ptr() + n * x.element_size(), d, d, n)
    return out


class DDPM(_HipBlock):
    def __init__(self, nn_model, betas, n_T, device, drop_prob=0.1):
        super().__init__()
        self.nn_model = nn_model.to(device)
        # the reference owns a GradScaler for its fp16 autocast (new_scripy.py:390); here it is live in float16 mode only
        # (fp32 / bf16 need no loss scaling) and keeps its state on the device (optim.DmGradScaler)
        from .optim import DmGradScaler
        self.scaler = DmGradScaler(enabled=self.nn_model.compute_dtype == torch.float16)
        for k, v in ddpm_schedules(betas[0], betas[1], n_T).items():
            self.register_buffer(k, v.to(device))
        self.n_T = n_T
        self.device = device
        self.drop_prob = drop_prob
        self.loss_mse = nn.MSELoss()
        self.n_classes = self.nn_model.n_classes
        self.compute_dtype = self.nn_model.compute_dtype
        self._cfg6 = (None, None)
        self._rng_calls = 0
        self._sample_calls = 0        # unseeded sample() calls so far: each draws from its own Philox key
        self.rng_seed = None          # None -> torch.initial_seed()

    # ------------------------------------------------------------------------------------------
    def _loss_constants(self, dev):
        vals = Cfg.loss_constants()               # read at call time, like the reference (new_scripy.py:420-435)
        if self._cfg6[0] != vals or self._cfg6[1].device != torch.device(dev):
            self._cfg6 = (list(vals), torch.tensor(vals, dtype=torch.float32, device=dev))
        return self._cfg6[1]

    def _seed(self):
        return int(torch.initial_seed() if self.rng_seed is None else self.rng_seed) & 0x7FFFFFFFFFFFFFFF

    def _sample_seed(self):
        """Philox key of the next UNSEEDED sample() call: the instance seed mixed (splitmix64) with a sampler domain tag and the
        number of unseeded calls so far.  The reference draws fresh torch.randn noise on every call (new_scripy.py:445,465), so two
        calls must not return the same images, and the sampler's stream must not coincide with the training noise, which uses the
        bare instance seed with the forward-call count as the counter's high word."""
        z = (self._seed() ^ 0x5A4D504C45520000) + 0x9E3779B97F4A7C15 * (self._sample_calls + 1)
        self._sample_calls += 1
        z &= 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return (z ^ (z >> 31)) & 0x7FFFFFFFFFFFFFFF

    def forward(self, x, c, attn_mask, *, ts=None, noise=None, ctx_mask=None):
        """Training loss (new_scripy.py:401-439).  x (B,3,H,W) fp32, c (B,) int, attn_mask (B,H,W)."""
        ops.L.require_device(x)
        dev = x.device
        B = x.shape[0]
        tfrac = None
        if ts is None or ctx_mask is None or noise is None:
            # the stream offset lives on the device and is advanced there, so that a captured train step (hipGraph or launch plan)
            # draws fresh timesteps, masks and noise on every replay
            if getattr(self, "_rng_dev", None) is None or self._rng_dev.device != dev:
                self._rng_dev = torch.full((1,), self._rng_calls, dtype=torch.int64, device=dev)
            self._rng_calls += 1
            ts_d = torch.empty(B, dtype=torch.int64, device=dev)
            tf_d, keep_d = torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.float32, device=dev)
            # t ~ U{1..n_T}, keep ~ Bernoulli(1 - drop_prob) (new_scripy.py:405, 413) by the library's Philox; advances the offset by one
            ops.call("dm_draw_ts_keep", ops.ptr(ts_d), ops.ptr(tf_d), ops.ptr(keep_d), B, int(self.n_T), float(1.0 - self.drop_prob),
                     self._seed(), ops.ptr(self._rng_dev))
            if ts is None:
                ts, tfrac = ts_d, tf_d
            if ctx_mask is None:
                ctx_mask = keep_d
            if noise is None:
                noise = ops.randn(tuple(x.shape), dev, self._seed(), self._rng_dev)
        ts = ts.to(dev).long()
        if tfrac is None:
            tfrac = ts.float() / self.n_T
        net = self.nn_model
        xt = ops.qsample(x.float(), noise, ts, self.sqrtab, self.sqrtmab, net.compute_dtype, _pad8(x.shape[1]))
        pred = net.decode(net._encode(xt), net.embed(c.to(dev), tfrac, ctx_mask.to(dev)))
        return ops.WeightedLoss.apply(pred, noise, attn_mask.to(dev).float().contiguous(), self._loss_constants(dev))

    # ------------------------------------------------------------------------------------------
    def _eps_cfg(self, x_i, c2, mask2, t2, ctx_embs, dedup):
        """eps for the doubled batch [mask 0 | mask 1] (2n,C,H,W)."""
        net = self.nn_model
        temb1, temb2 = net.time_emb1(t2.reshape(-1, 1)), net.time_emb2(t2.reshape(-1, 1))
        embs = (ctx_embs[0], temb1, ctx_embs[1], temb2)
        if dedup and not net.training:
            # encoder + up0 once on n (exact in eval mode); the skip tensors stay at batch n — the decoder's upsample-concat and the
            # head convolution read sample b % n (no doubled copies: x0 alone is 67 MB) — only the small bottleneck map is doubled
            x0, d1, d2, d3, d4, u1 = net.encode(x_i)
            F, (n, H, W) = x0.shape[3], x0.shape[:3]
            bc = ops.conv_bcast_ok(x0.dtype, 2 * n, H, W, F, F)      # the head conv's broadcast second source needs the halo-resident kernel
            feats = (x0 if bc else repeat2(x0), d1, d2, d3, d4, repeat2(u1))
        else:
            feats = net.encode(repeat2(x_i))
        return net.decode(feats, embs)

    @torch.no_grad()
    def sample(self, n_sample, size, device, guide_w=0.0, refine_steps=2, *, x_T=None, zs=None, dedup=True,
               use_graph=False, seed=None, steps=None, verbose=False, first_sample=0, total_samples=None):
        """Ancestral sampling with doubled-batch CFG (new_scripy.py:441-477).  `refine_steps` is unused, as in the
        reference.  The guidance convention is the reference's: the first half of the doubled batch has
        ctx_mask 0, so eps = (1+w)*eps_uncond - w*eps_cond.

        x_T / zs inject the initial noise and the per-step noise (zs[j] is used at step i = n_T - j);
        otherwise Philox noise is generated in-kernel.  use_graph captures one step as a hipGraph and
        replays it (requires in-kernel noise).  `steps` stops after that many iterations (tests).

        first_sample / total_samples (parallel.sample_sharded): this call produces samples [first_sample, first_sample +
        n_sample) of a class-cycled batch of total_samples — their classes and their slice of the in-kernel noise
        stream — so the shards of any world size concatenate to the images of the single-process call with that seed."""
        net = self.nn_model
        ops.L.ensure_workspace()
        ops.L.device_guard(recheck=True)             # one process per GPU (the halo kernels' LDS does not survive preemption between processes)
        total = n_sample if total_samples is None else int(total_samples)
        if total % self.n_classes:
            raise DmError(f"n_sample={total} must be a multiple of n_classes={self.n_classes} (new_scripy.py:448)")
        if first_sample < 0 or first_sample + n_sample > total:
            raise DmError(f"samples [{first_sample}, {first_sample + n_sample}) are not inside a batch of {total}")
        dev = torch.device(device)
        seed = self._sample_seed() if seed is None else int(seed)
        per = 1
        for d in size:
            per *= int(d)
        first_elem = first_sample * per
        if first_elem % 4:
            raise DmError(f"a shard must start on a multiple of 4 elements (first_sample={first_sample}, {per} elements per sample)")
        x_i = (x_T.to(dev).float().contiguous().clone() if x_T is not None
               else ops.randn((n_sample,) + tuple(size), dev, seed, 0, first_elem))
        c_i = ((first_sample + torch.arange(0, n_sample, device=dev)) % self.n_classes).repeat(2)      # arange(n_classes).repeat(..) (:448)
        mask = torch.zeros(2 * n_sample, device=dev)
        mask[n_sample:] = 1.0
        oh = ops.onehot_mask(c_i, mask, self.n_classes)
        ctx_embs = (net.ctx_emb1(oh), net.ctx_emb2(oh))               # constant over the trajectory
        step = torch.full((1,), self.n_T, dtype=torch.int32, device=dev)
        t2 = torch.empty(2 * n_sample, dtype=torch.float32, device=dev)
        sched = {k: getattr(self, k) for k in ("oneover_sqrta", "mab_over_sqrtmab", "sqrt_beta_t")}
        n_iter = self.n_T if steps is None else steps

        def one_step(z):
            ops.fill_t(t2, step, self.n_T)
            eps = self._eps_cfg(x_i, c_i, mask, t2, ctx_embs, dedup)
            ops.cfg_update(x_i, eps, z, guide_w, sched, step, seed=seed, dec_step=True, first_elem=first_elem)

        if use_graph:
            if zs is not None:
                raise DmError("use_graph needs in-kernel noise (zs=None)")
            one_step(None)                                            # warm-up: fills packed-weight / BN-fold caches
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):      # (the RCCL watchdog thread may poll events meanwhile)
                one_step(None)
            for _ in range(n_iter - 1):
                g.replay()
        else:
            for j in range(n_iter):
                if verbose:
                    print(f"sampling step {self.n_T - j}", end="\r")
                z = None
                if zs is not None and self.n_T - j > 1:
                    z = zs[j].to(dev).float().contiguous()
                one_step(z)
        return x_i
