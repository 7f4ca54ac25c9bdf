"""One hipGraph for the whole train step (zero_grad, loss forward, backward, clip + AdamW, weight re-packs).

The eager step issues ~1300 kernel launches from Python; at ~12 us of host work per launch that is within
15 % of the GPU time of the step, so the host starts to pace the device.  Every launch of the step goes to
torch's current stream (the C ABI takes the stream as an argument) and nothing in it needs the host:
the timesteps / context mask come from torch's device RNG, the DDPM noise from the Philox kernel with a
device-resident stream offset, the AdamW bias corrections from a device-resident step count.  So the step
is captured once and replayed.

    step = GraphedTrainStep(ddpm, opt, x, c, attn_mask)     # warm-up + capture; model / optimiser state is left untouched
    loss = step(x, c, attn_mask)                             # copies the batch in, replays, returns the (static) loss tensor

Constraints: fixed batch shape; single process (no collective inside the capture — use the eager step
with GradReducer for data parallelism); hyper-parameters are read from the param group between replays.
"""
import ctypes as C

import torch

from . import _lib as L
from . import ops
from ._lib import DmError
from .modules import _HipBlock


class LaunchPlan:
    """The kernel / memset nodes of a captured (not instantiated) hipGraph, re-issued from C as plain launches on the current
    stream (include/dm_amd.h, dm_plan_*).  Holds the torch CUDAGraph: its nodes own the argument blocks and its private memory
    pool owns every tensor the step allocates."""

    def __init__(self, cuda_graph):
        lib = L.load()
        self._graph = cuda_graph
        raw = cuda_graph.raw_cuda_graph()
        h = C.c_void_p()
        if lib.dm_plan_from_graph(C.c_void_p(int(raw)), C.byref(h)) != 0:
            raise DmError("launch plan: " + lib.dm_last_error().decode())
        self._h = h
        info = (C.c_int32 * 6)()
        lib.dm_plan_info(h, info)
        self.n_ops, self.n_kernels, self.n_memsets, self.n_markers, self.n_skipped, self.n_segments = [int(v) for v in info]
        self.segment_markers = [int(lib.dm_plan_segment_marker(h, s)) for s in range(self.n_segments)]

    def run(self, first=0, last=None):
        L.call("dm_plan_run", self._h, int(first), int(self.n_segments - 1 if last is None else last))

    def run_timed(self, substr, first=0, last=None):
        """run() with a HIP event pair around every kernel whose name contains one of the '|'-separated substrings of `substr`
        ("" = every kernel); read them with timed_results()."""
        L.call("dm_plan_run_timed", self._h, int(first), int(self.n_segments - 1 if last is None else last), substr.encode())

    def timed_results(self, cap=8192):
        """[(op index, kernel name, ms)] of the launches timed since the last call (synchronises the stream)."""
        ms, op, n = (C.c_float * cap)(), (C.c_int32 * cap)(), C.c_int32(0)
        L.call("dm_plan_timed_results", self._h, C.cast(ms, C.c_void_p), C.cast(op, C.c_void_p), cap, C.cast(C.byref(n), C.c_void_p))
        names = self.op_names()
        return [(int(op[i]), names[op[i]][1], float(ms[i])) for i in range(n.value)]

    def op_names(self):
        lib = L.load()
        buf = C.create_string_buffer(512)
        out = []
        for i in range(self.n_ops):
            kind = lib.dm_plan_op_name(self._h, i, buf, 512)
            out.append((kind, buf.value.decode(errors="replace")))
        return out

    def foreign_kernels(self):
        """Kernels in the plan that are not libdm_amd's — torch-issued elementwise / fill / copy / RNG kernels (mangled `at::…` names)
        or runtime copy kernels: {name: count}.  A planned train step holds a handful (fills of the gradient buffers, the step count)."""
        tally = {}
        for kind, nm in self.op_names():
            if kind == 0 and ("_ZN2at" in nm or "at::" in nm or "rocclr" in nm):
                tally[nm] = tally.get(nm, 0) + 1
        return tally

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.load().dm_plan_destroy(self._h)
        except Exception:                      # noqa: BLE001 — interpreter shutdown
            pass


class GraphedTrainStep:
    """mode="graph": replay the instantiated hipGraph.  mode="plan": the captured graph is never instantiated; its kernel /
    memset nodes are read back once (dm_plan_from_graph) and every step re-issues them from C as plain stream launches
    (dm_plan_run: one ctypes call per step) — the host cost of a graph replay without the graph executor's per-node device
    cost, and segments (dm_plan_marker) between which the host can run collectives (parallel.OverlappedGradReducer.replay)."""

    def __init__(self, ddpm, opt, x, c, attn_mask, warmup=2, mode="graph", body=None, runner=None):
        if not x.is_cuda:
            raise DmError("GraphedTrainStep needs device tensors")
        if mode not in ("graph", "plan"):
            raise DmError(f"GraphedTrainStep: mode must be 'graph' or 'plan', got {mode!r}")
        if runner is not None and mode != "plan":
            raise DmError("GraphedTrainStep: a segment runner needs mode='plan'")
        self.ddpm, self.opt, self.mode, self._body, self._runner, self.plan = ddpm, opt, mode, body, runner, None
        L.ensure_workspace()
        self._shared_at_capture, self._fallback_eager = L.device_guard(recheck=True), False
        self.x, self.c, self.am = x.clone(), c.clone(), attn_mask.clone()
        self._specs = [sp for m in ddpm.modules() if isinstance(m, _HipBlock) for sp in m._specs()]
        dev = x.device
        if getattr(ddpm, "_rng_dev", None) is None or ddpm._rng_dev.device != dev:
            ddpm._rng_dev = torch.full((1,), ddpm._rng_calls, dtype=torch.int64, device=dev)

        # ---- everything the warm-up steps change is put back afterwards
        snap = dict(p=opt.flat_p.clone(), m=opt.exp_avg.clone(), v=opt.exp_avg_sq.clone(), t=opt._step_dev.clone(),
                    rng=ddpm._rng_dev.clone(), bufs=[b.clone() for b in ddpm.buffers()], torch_rng=torch.cuda.get_rng_state(dev),
                    step=opt._step, calls=ddpm._rng_calls, nbt=[sp.nbt_pending for sp in self._specs])
        scaler = getattr(ddpm, "scaler", None)
        scal_state = scaler._state_on(dev).clone() if (scaler is not None and scaler.is_enabled()) else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # allocator and cache warm-up off the capture stream
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        before = [sp.nbt_pending for sp in self._specs]
        self.graph = torch.cuda.CUDAGraph(keep_graph=True) if mode == "plan" else torch.cuda.CUDAGraph()
        ops.PROFILE_META = meta = []               # (kernel family, algorithmic FLOPs, shape) of every MFMA launch, in launch order
        try:
            # thread_local: other threads of the process (the RCCL watchdog polls its events) may keep calling the runtime during the capture
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.loss = self._eager()
        finally:
            ops.PROFILE_META = None
        self.conv_meta = meta
        if mode == "plan":
            self.plan = LaunchPlan(self.graph)
        self._nbt = [sp.nbt_pending - b for sp, b in zip(self._specs, before)]     # BatchNorm batch counters of one step
        with torch.no_grad():
            opt.flat_p.copy_(snap["p"]); opt.exp_avg.copy_(snap["m"]); opt.exp_avg_sq.copy_(snap["v"])
            opt._step_dev.copy_(snap["t"]); ddpm._rng_dev.copy_(snap["rng"])
            for b, s in zip(ddpm.buffers(), snap["bufs"]):
                b.copy_(s)
            if scal_state is not None:
                scaler._state_on(dev).copy_(scal_state)
        torch.cuda.set_rng_state(snap["torch_rng"], dev)
        opt._step, ddpm._rng_calls = snap["step"], snap["calls"]
        for sp, n in zip(self._specs, snap["nbt"]):
            sp.nbt_pending = n
        opt.flat_g.zero_()
        opt.refresh_shadow()
        ops.bump_weight_epoch()
        ops.refresh_packs()
        self.replays = 0

    def _eager(self):
        if self._body is not None:
            return self._body(self)
        self.opt.zero_grad()
        loss = self.ddpm(self.x, self.c, self.am)
        scaler = getattr(self.ddpm, "scaler", None)
        if scaler is not None and scaler.is_enabled():         # float16 mode: new_scripy.py:792-801
            scaler.scale(loss).backward()
            scaler.step(self.opt)
        else:
            loss.backward()
            self.opt.step()
        return loss

    def _replay(self):
        if self._runner is not None:
            self._runner(self.plan)
        elif self.plan is not None:
            self.plan.run()
        else:
            self.graph.replay()

    def __call__(self, x=None, c=None, attn_mask=None):
        if not self._shared_at_capture and L.device_guard(recheck=True):
            # another process started using this GPU: the captured launches hold > 64 KiB of LDS per workgroup, which is not
            # preemption-safe between processes here (_lib.device_guard) -> from now on the step is issued eagerly, on the small kernels
            self._fallback_eager = True
        if self._fallback_eager:
            if x is not None:
                self.x.copy_(x, non_blocking=True)
            if c is not None:
                self.c.copy_(c, non_blocking=True)
            if attn_mask is not None:
                self.am.copy_(attn_mask, non_blocking=True)
            return self._eager()
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if c is not None:
            self.c.copy_(c, non_blocking=True)
        if attn_mask is not None:
            self.am.copy_(attn_mask, non_blocking=True)
        self.opt.sync_hyper()                              # picks up a changed lr / weight decay (outside the graph)
        if ops.ZERO_ARENA.off:
            # an eager train-mode forward / backward that was not followed by opt.step() (a logging pass, a partial accumulation
            # group) left non-zero sums in arena slices the captured step will use as zeroed accumulators: the replay zeroes
            # the arena only at its END, so put it back in the state the capture assumed
            ops.ZERO_ARENA.recycle()
        self._replay()
        for sp, d in zip(self._specs, self._nbt):          # host-side bookkeeping one replay stands for
            sp.nbt_pending += d
        self.opt._step += 1
        self.ddpm._rng_calls += 1
        self.replays += 1
        # (packed weights: every pack is either refreshed by the captured optimiser step (bf16 shadow, registered transposes) or
        #  was stale at capture time and is therefore rebuilt inside the graph, so eager code that follows finds fresh contents
        #  behind the stamps the capture left)
        return self.loss
