"""Fused clip_grad_norm_ + AdamW on flat fp32 buffers (reference: new_scripy.py:715-719, 797-803).

All parameters are re-pointed at slices of one flat fp32 buffer (4-D weights keep their channels_last
physical order), the gradients live in a second flat buffer: the conv weight-gradient kernels
accumulate straight into it (`param.main_grad`), the few hundred small parameters' autograd grads
are gathered by ONE multi-tensor launch, and one sum-of-squares + one AdamW launch finish the step.
The flat gradient buffer is also what the data-parallel all-reduce works on (parallel.py).
"""
import os
import weakref

import torch

from . import ops
from ._lib import DmError, call, ptr


class FusedAdamW(torch.optim.Optimizer):
    CHUNK = 16384

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, max_grad_norm=1.0,
                 grad_scale=1.0, shadow_dtype=torch.bfloat16):
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
        if shadow_dtype not in (torch.bfloat16, torch.float16):
            raise DmError(f"FusedAdamW: the 16-bit parameter shadow is bfloat16 or float16 (the model's compute dtype), got {shadow_dtype}")
        self.flat_p16 = torch.zeros(total, dtype=shadow_dtype, device=dev)       # 16-bit shadow, refreshed by the AdamW kernel
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
        ops._T_TABLES.clear()                       # pack tables built before the shadow existed point at the fp32 masters
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._hyper = torch.zeros(9, dtype=torch.float32, device=dev)
        self._hyper_host = None                       # last values uploaded: re-sent only when a hyper-parameter changes
        self._table = None                            # (pinned host, device) pointer table of gather_grads
        self._step_dev = torch.zeros(1, dtype=torch.int64, device=dev)   # step count on the device (bias corrections)
        self._lazy = None                             # (table, rows, parameters) of the lazy zero_grad, learned from the first step
        self._lazy_enabled = os.environ.get("DM_LAZY_ZERO", "1") != "0"
        self._captured_tables = []                    # pointer tables a stream capture baked into a graph: never reused
        self._capture_spare = None                    # the table the next capture will take (allocated on the eager path)
        self.refresh_shadow()
        ops.bump_weight_epoch()
        # the gradient scratch arena is recycled by step(); that is only sound while ONE optimiser consumes the gradients of
        # this process (a second live optimiser's step would zero accumulators the first one has not read yet)
        _LIVE.add(self)
        ops.ARENA_ENABLED[0] = len(_LIVE) == 1
        if not ops.ARENA_ENABLED[0]:
            ops.ZERO_ARENA.recycle()

    def refresh_shadow(self):
        """Recompute the whole bf16 shadow from the fp32 masters (construction, after a parameter broadcast or any
        other write to `flat_p` that did not go through step())."""
        call("dm_cast", ptr(self.flat_p), ptr(self.flat_p16), ops.L.DM_F32, ops.dt(self.flat_p16), self.total)
        for p, _, _ in self._slots:
            if hasattr(p, "_dm_shadow16"):
                p._dm_shadow_stamp = (p.data_ptr(), p._version)

    # ------------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none=True):
        """r04: the ranges of the parameters whose weight gradient comes from the halo kernel + its reduce launch (the 3x3 layers: ~3/4 of
        the buffer) are NOT zeroed — they are marked fresh and their first weight-gradient launch overwrites (DmWgrad.overwrite: the
        reduce launch writes instead of read-modify-writing): the 426-MB fill becomes one dm_zero_ranges launch over the rest.  Which
        parameters those are is learned from the first complete step (ops sets `_dm_halo_ok`); until then, and with DM_LAZY_ZERO=0, the
        whole buffer is filled.  A fresh parameter that no launch wrote is zeroed by settle_fresh() before anything reads the buffer."""
        if self._lazy is None:
            self.flat_g.zero_()
        else:
            table, n_rows, lazy = self._lazy
            if n_rows:
                call("dm_zero_ranges", ptr(table), n_rows)
            for p in lazy:
                p._dm_fresh = True
        for p, _, _ in self._slots:
            p.grad = None

    def settle_fresh(self):
        """Zero the gradient range of every lazily-zeroed parameter that no weight-gradient launch has written since zero_grad()
        (an unused layer, a rank without a micro-batch): called before anything reads the flat buffer — the all-reduce, the norm, AdamW."""
        if self._lazy is None:
            return
        for p in self._lazy[2]:
            if getattr(p, "_dm_fresh", False):
                p.main_grad.zero_()
                p._dm_fresh = False

    def _learn_lazy(self):
        """After the first complete step: the parameters every weight-gradient launch of which took the halo kernel -> the lazy set; the
        complement of their ranges -> the table dm_zero_ranges walks (chunks of <= 256 Ki floats, one workgroup each)."""
        lazy = [p for p, _, _ in self._slots if hasattr(p, "main_grad") and getattr(p, "_dm_halo_ok", False)]
        ids = {id(p) for p in lazy}
        rows, base, lo = [], self.flat_g.data_ptr(), None
        spans, cur = [], None
        for p, off, n in self._slots:
            n4 = (n + 3) // 4 * 4
            if id(p) in ids:
                if cur is not None:
                    spans.append(cur)
                    cur = None
            else:
                cur = [off, off + n4] if cur is None else [cur[0], off + n4]
        if cur is not None:
            spans.append(cur)
        for a, b in spans:
            for x in range(a, b, 1 << 18):
                rows.append((base + 4 * x, min(1 << 18, b - x)))
        table = torch.tensor(rows if rows else [(0, 0)], dtype=torch.int64).to(self.flat_g.device)
        self._lazy = (table, len(rows), lazy)

    def _new_table(self, cap):
        return [torch.empty((cap, 3), dtype=torch.int64).pin_memory(),
                torch.empty((cap, 3), dtype=torch.int64, device=self.flat_g.device), None]

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
            # pointer table: pinned host buffers (so the upload is a plain async copy, legal in a stream capture), a ring of
            # four guarded by events because the host runs ahead of the device in eager mode
            capturing = torch.cuda.is_current_stream_capturing()
            if not capturing and (self._table is None or self._table[0][0].shape[0] < len(rows)):
                cap = max(2 * len(rows), 1024)
                self._table = [self._new_table(cap) for _ in range(4)]
                self._table_i = 0
            if capturing:
                # the upload becomes a graph node that re-reads its pinned source on every replay: the capture gets a table of its
                # own (host and device) that lives as long as the optimiser and is never recycled by later eager steps.  Pinned
                # memory cannot be allocated inside a capture, so the spare was set aside by the last eager call.
                slot = self._capture_spare
                if slot is None or slot[0].shape[0] < len(rows):
                    raise DmError("FusedAdamW: run one eager step() before capturing it in a graph (the pointer table of the "
                                  "gradient gather is allocated on the eager path)")
                self._capture_spare = None
                self._captured_tables.append(slot)
            else:
                if self._capture_spare is None or self._capture_spare[0].shape[0] < len(rows):
                    self._capture_spare = self._new_table(max(2 * len(rows), 1024))
                slot = self._table[self._table_i]
                self._table_i = (self._table_i + 1) % 4
                if slot[2] is not None:
                    slot[2].synchronize()             # the copy that last read this host buffer has run
            host, table = slot[0], slot[1]
            host[:len(rows)] = torch.tensor(rows, dtype=torch.int64)
            if capturing:
                # a library copy kernel reads the pinned host table directly (a torch copy_ would become a memcpy node, which
                # a launch plan cannot replay: dm_plan_from_graph)
                call("dm_cast", host.data_ptr(), table.data_ptr(), ops.L.DM_F32, ops.L.DM_F32, 6 * len(rows))
            else:
                table[:len(rows)].copy_(host[:len(rows)], non_blocking=True)
            if not capturing:
                slot[2] = torch.cuda.Event()
                slot[2].record()
            call("dm_scatter_copy", ptr(table), len(rows), 1)
        for p, _, _ in self._slots:
            p.grad = None
        if ops.ARENA_ENABLED[0]:
            ops.ZERO_ARENA.recycle()                 # every small gradient accumulator has been copied out: one fill re-zeroes them

    @torch.no_grad()
    def step(self, closure=None, scaler=None):
        """clip_grad_norm_(max_grad_norm) + AdamW (new_scripy.py:797-801).  `scaler` (DmGradScaler, fp16 mode): the gradients carry
        the loss scale; the kernel unscales them, a step with inf / nan gradients is skipped and the scale adapts — all on the device."""
        if closure is not None:
            raise DmError("FusedAdamW.step does not take a closure")
        ops.L.device_guard(recheck=True)             # one process per GPU, re-tested every step (two system calls)
        self.settle_fresh()
        self.gather_grads()
        self._step += 1
        self.sync_hyper()
        self._step_dev.add_(1)                       # the kernel derives 1 - beta^t from the device-side count
        self._sumsq.zero_()
        call("dm_sumsq", ptr(self.flat_g), self.total, ptr(self._sumsq))
        state = None
        if scaler is not None and scaler.is_enabled():
            state = scaler._state_on(self.flat_p.device)
            self._scaled = True                      # a skipped step takes the device-side count back: the host count is then only an upper bound
            call("dm_scaler_update", ptr(state), ptr(self._sumsq), ptr(self._step_dev), float(scaler._growth_factor), float(scaler._backoff_factor),
                 int(scaler._growth_interval))
        call("dm_adamw_scaled", ptr(self.flat_p), ptr(self.flat_g), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.total,
             ptr(self._sumsq), ptr(self._hyper), ptr(self.flat_p16), ops.dt(self.flat_p16), ptr(self._step_dev), ptr(state))
        ops.bump_weight_epoch()
        ops.refresh_packs()                          # every transposed (input-gradient) weight pack, one launch
        if self._lazy is None and self._lazy_enabled and not torch.cuda.is_current_stream_capturing():
            self._learn_lazy()

    def sync_hyper(self):
        """Upload lr / betas / eps / weight decay / clip norm when they differ from the device copy (lr schedules)."""
        g0 = self.param_groups[0]
        b1, b2 = g0["betas"]
        hyper = [g0["lr"], b1, b2, g0["eps"], g0["weight_decay"], self.max_grad_norm, self.grad_scale, 0.0, 0.0]
        if hyper != self._hyper_host:
            if torch.cuda.is_current_stream_capturing():
                raise DmError("FusedAdamW: a hyper-parameter changed inside a stream capture; change it between replays")
            self._hyper.copy_(torch.tensor(hyper, dtype=torch.float32))
            self._hyper_host = hyper

    def grad_norm(self):
        """Global gradient norm of the last step (device tensor, no host sync)."""
        return self._sumsq.sqrt() * self.grad_scale

    # ------------------------------------------------------------------------------------------
    # torch.optim.AdamW's state_dict() schema (what the reference checkpoints hold: new_scripy.py:736-743 saves optim.state_dict())
    _GROUP_DEFAULTS = dict(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                           decoupled_weight_decay=True)

    def _logical(self, flat, p, off, n):
        """The slice of a flat state buffer that belongs to `p`, as a fresh tensor in p's logical shape."""
        if p.dim() == 4:
            O, I, kh, kw = p.shape
            return flat[off:off + n].view(O, kh, kw, I).permute(0, 3, 1, 2).contiguous()
        return flat[off:off + n].view(p.shape).clone()

    def state_dict(self):
        """{"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]} exactly as torch.optim.AdamW lays it out
        (per-parameter tensors in the parameters' logical shapes, `step` a float32 scalar tensor, empty state before the first
        step), so a checkpoint written here resumes under torch's AdamW and vice versa."""
        g0 = self.param_groups[0]
        group = {k: g0[k] for k in ("lr", "betas", "eps", "weight_decay")}
        for k, v in self._GROUP_DEFAULTS.items():
            group[k] = g0.get(k, v)
        for k in g0:                                   # e.g. initial_lr written by an lr scheduler
            if k not in group and k != "params":
                group[k] = g0[k]
        group["params"] = list(range(len(self._slots)))
        state = {}
        if getattr(self, "_scaled", False):            # loss scaling skipped steps on the device: that count is the truth
            self._step = int(self._step_dev.item())
        if self._step > 0:
            for i, (p, off, n) in enumerate(self._slots):
                state[i] = {"step": torch.tensor(float(self._step), dtype=torch.float32),
                            "exp_avg": self._logical(self.exp_avg, p, off, n),
                            "exp_avg_sq": self._logical(self.exp_avg_sq, p, off, n)}
        return {"state": state, "param_groups": [group]}

    @torch.no_grad()
    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self._slots):
            raise DmError(f"FusedAdamW.load_state_dict: expected one param group of {len(self._slots)} parameters, got "
                          f"{[len(g['params']) for g in groups]}")
        g0 = self.param_groups[0]
        for k, v in groups[0].items():
            if k != "params":
                g0[k] = tuple(v) if k == "betas" else v
        state = sd.get("state", {})
        steps = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, (p, off, n) in enumerate(self._slots):
            st = state.get(i, state.get(str(i)))
            if st is None:
                continue
            steps.add(int(float(st["step"])))
            for key, flat in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
                v = st[key]
                if tuple(v.shape) != tuple(p.shape):
                    raise DmError(f"FusedAdamW.load_state_dict: {key} of parameter {i} has shape {tuple(v.shape)}, expected {tuple(p.shape)}")
                v = v.to(device=flat.device, dtype=torch.float32)
                dst = flat[off:off + n]
                if p.dim() == 4:
                    O, I, kh, kw = p.shape
                    dst.view(O, kh, kw, I).copy_(v.permute(0, 2, 3, 1))
                else:
                    dst.view(p.shape).copy_(v)
        if len(steps) > 1:
            raise DmError(f"FusedAdamW.load_state_dict: parameters disagree on the step count ({sorted(steps)}); the fused kernel keeps one")
        self._step = steps.pop() if steps else 0
        self._step_dev.fill_(self._step)
        self._hyper_host = None
        self.refresh_shadow()                          # the masters were most likely just loaded too
        ops.bump_weight_epoch()
        ops.refresh_packs()


class DmGradScaler:
    """torch.amp.GradScaler's surface (the reference owns one as `ddpm.scaler` and drives it at new_scripy.py:792-802:
    scale(loss).backward(); unscale_(optim); clip_grad_norm_; step(optim); update()) with the state on the DEVICE, so the whole
    train step stays capturable: {scale, growth tracker, found_inf, 1/scale} live in a 4-float tensor that dm_scaler_update /
    dm_adamw_scaled read and write.  Defaults are torch's: init 2^16, growth 2.0, backoff 0.5, interval 2000.
    Only FusedAdamW can be stepped through it (its kernel unscales, clips and skips); enabled=False makes every call a pass-through."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self._enabled = bool(enabled)
        self._init_scale, self._growth_factor, self._backoff_factor, self._growth_interval = float(init_scale), float(growth_factor), float(backoff_factor), int(growth_interval)
        self._state = None

    def is_enabled(self):
        return self._enabled

    def _state_on(self, device):
        if self._state is None or self._state.device != torch.device(device):
            prev = self._state.cpu() if self._state is not None else torch.tensor([self._init_scale, 0.0, 0.0, 1.0 / self._init_scale])
            self._state = prev.to(device=device, dtype=torch.float32)
        return self._state

    def scale(self, outputs):
        if not self._enabled:
            return outputs
        st = self._state_on(outputs.device)
        return outputs * st[0]

    def unscale_(self, optimizer):
        """The fused kernel unscales (and clips the unscaled gradient) inside step(); kept for the reference's call sequence."""
        if self._enabled and not isinstance(optimizer, FusedAdamW):
            raise DmError("DmGradScaler drives FusedAdamW only (its kernel unscales, clips and skips on the device)")

    def step(self, optimizer, *args, **kwargs):
        if not self._enabled:
            return optimizer.step(*args, **kwargs)
        if not isinstance(optimizer, FusedAdamW):
            raise DmError("DmGradScaler drives FusedAdamW only (its kernel unscales, clips and skips on the device)")
        return optimizer.step(scaler=self)

    def update(self, new_scale=None):
        """The scale was already adapted on the device by step(); an explicit new_scale overrides it."""
        if self._enabled and new_scale is not None:
            st = self._state_on(self._state.device if self._state is not None else "cuda")
            st[0] = float(new_scale)
            st[1] = 0.0

    def get_scale(self):
        if not self._enabled:
            return 1.0
        return float(self._state[0].item()) if self._state is not None else self._init_scale

    def found_inf_last_step(self):
        return bool(self._state is not None and self._state[2].item() != 0.0)

    def state_dict(self):
        if not self._enabled:
            return {}
        tracker = int(self._state[1].item()) if self._state is not None else 0
        return {"scale": self.get_scale(), "growth_factor": self._growth_factor, "backoff_factor": self._backoff_factor,
                "growth_interval": self._growth_interval, "_growth_tracker": tracker}

    def load_state_dict(self, sd):
        if not self._enabled or not sd:
            return
        self._growth_factor, self._backoff_factor, self._growth_interval = float(sd["growth_factor"]), float(sd["backoff_factor"]), int(sd["growth_interval"])
        dev = self._state.device if self._state is not None else None
        self._state = torch.tensor([float(sd["scale"]), float(sd.get("_growth_tracker", 0)), 0.0, 1.0 / float(sd["scale"])])
        if dev is not None:
            self._state = self._state.to(dev)


_LIVE = weakref.WeakSet()       # optimisers alive in this process (the gradient scratch arena serves exactly one)
