"""Fused clip_grad_norm_ + AdamW on flat fp32 buffers (reference: new_scripy.py:715-719, 797-803).

All parameters are re-pointed at slices of one flat fp32 buffer (4-D weights keep their channels_last
physical order), the gradients live in a second flat buffer: the conv weight-gradient kernels
accumulate straight into it (`param.main_grad`), the few hundred small parameters' autograd grads
are gathered by ONE multi-tensor launch, and one sum-of-squares + one AdamW launch finish the step.
The flat gradient buffer is also what the data-parallel all-reduce works on (parallel.py).
"""
import math

import torch

from . import ops
from ._lib import DmError, call, ptr


class FusedAdamW(torch.optim.Optimizer):
    CHUNK = 16384

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, max_grad_norm=1.0,
                 grad_scale=1.0):
        params = [p for p in params]
        if not params:
            raise DmError("FusedAdamW: no parameters")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.max_grad_norm = 0.0 if max_grad_norm is None else float(max_grad_norm)
        self.grad_scale = float(grad_scale)       # e.g. 1/world_size after a sum all-reduce
        self._step = 0
        ps = [p for g in self.param_groups for p in g["params"]]
        dev = ps[0].device
        ops.L.require_device(*ps)
        offs, total = [], 0
        for p in ps:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4       # keep every slice 16-byte aligned
        self.total = total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_p16 = torch.zeros(total, dtype=torch.bfloat16, device=dev)     # bf16 shadow, refreshed by the AdamW kernel
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self._slots = []
        with torch.no_grad():
            for p, off in zip(ps, offs):
                n = p.numel()
                if p.dim() == 4:
                    ops._cl(p)
                    O, I, kh, kw = p.shape
                    phys = self.flat_p[off:off + n].view(O, kh, kw, I)
                    phys.copy_(p.data.permute(0, 2, 3, 1))
                    p.data = phys.permute(0, 3, 1, 2)
                    p.main_grad = self.flat_g[off:off + n].view(O, kh, kw, I)
                    p._dm_shadow16 = self.flat_p16[off:off + n].view(O, kh * kw, I)
                    p._dm_shadow_stamp = None
                else:
                    flat = self.flat_p[off:off + n].view(p.shape)
                    flat.copy_(p.data)
                    p.data = flat
                self._slots.append((p, off, n))
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._hyper = torch.zeros(9, dtype=torch.float32, device=dev)
        self.refresh_shadow()
        ops.bump_weight_epoch()
        ops.ARENA_ENABLED[0] = True                  # step() recycles the gradient scratch arena

    def refresh_shadow(self):
        """Recompute the whole bf16 shadow from the fp32 masters (construction, after a parameter broadcast or any
        other write to `flat_p` that did not go through step())."""
        call("dm_cast", ptr(self.flat_p), ptr(self.flat_p16), ops.L.DM_F32, ops.L.DM_BF16, self.total)
        for p, _, _ in self._slots:
            if hasattr(p, "_dm_shadow16"):
                p._dm_shadow_stamp = (p.data_ptr(), p._version)

    # ------------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none=True):
        self.flat_g.zero_()
        for p, _, _ in self._slots:
            p.grad = None

    def gather_grads(self):
        """Fold autograd-produced `.grad` tensors (everything that is not a main_grad conv weight)
        into the flat gradient buffer with one multi-tensor launch."""
        rows = []
        base = self.flat_g.data_ptr()
        for p, off, n in self._slots:
            g = p.grad
            if g is None:
                continue
            if p.dim() == 4:
                g = g.permute(0, 2, 3, 1)
            if not g.is_contiguous():
                g = g.contiguous()
            p._keep = g                              # keep alive until the kernel has run
            for lo in range(0, n, self.CHUNK):       # one workgroup per <= CHUNK elements
                rows.append((g.data_ptr() + 4 * lo, base + 4 * (off + lo), min(self.CHUNK, n - lo)))
        if rows:
            table = torch.tensor(rows, dtype=torch.int64).to(self.flat_g.device, non_blocking=True)
            call("dm_scatter_copy", ptr(table), len(rows), 1)
            self._table = table
        for p, _, _ in self._slots:
            p.grad = None
        ops.ZERO_ARENA.recycle()                     # every small gradient accumulator has been copied out: one fill re-zeroes them

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise DmError("FusedAdamW.step does not take a closure")
        self.gather_grads()
        g0 = self.param_groups[0]
        b1, b2 = g0["betas"]
        self._step += 1
        hyper = [g0["lr"], b1, b2, g0["eps"], g0["weight_decay"], self.max_grad_norm, self.grad_scale,
                 1.0 - b1 ** self._step, 1.0 - b2 ** self._step]
        self._hyper.copy_(torch.tensor(hyper, dtype=torch.float32), non_blocking=True)
        self._sumsq.zero_()
        call("dm_sumsq", ptr(self.flat_g), self.total, ptr(self._sumsq))
        call("dm_adamw", ptr(self.flat_p), ptr(self.flat_g), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.total,
             ptr(self._sumsq), ptr(self._hyper), ptr(self.flat_p16))
        ops.bump_weight_epoch()
        ops.refresh_packs()                          # every transposed (input-gradient) weight pack, one launch

    def grad_norm(self):
        """Global gradient norm of the last step (device tensor, no host sync)."""
        return self._sumsq.sqrt() * self.grad_scale
