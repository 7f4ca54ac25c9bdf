"""Batch data parallelism: one process per GPU, RCCL all-reduce of the flat gradient buffer over xGMI.

The reference has no distributed code; its gradient-accumulation semantics define the target
(new_scripy.py:786, 795-803): every rank is one micro-batch with LOCAL BatchNorm statistics, the
gradients are summed and divided by the number of micro-batches, clipping acts on the averaged
gradient.  So rank == micro-batch, and `ACCUM_STEPS = world_size` on one device is the oracle.

xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU), so one big ring all-reduce is bound by a
single link; the flat buffer is therefore cut into a few large buckets that RCCL can pipeline, each
issued on a side stream as soon as the backward pass has produced it (reverse registration order),
and the optimiser waits on the side stream only right before its first kernel.
"""
import os

import torch
import torch.distributed as dist

from ._lib import DmError


def init_from_env(backend=None, force=False):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if dist.is_initialized() and dist.get_world_size() > 1:
        guard_shared_device(local)
    return rank, world, local


def _device_identity(local):
    """(host, physical device) of the HIP device this rank will use — enough to tell whether two ranks share one GPU."""
    import socket
    from ._lib import device_identity
    return socket.gethostname(), device_identity(local)


def guard_shared_device(local, group=None, identity=None):
    """Two ranks on ONE GPU time-share it, and on this driver stack a workgroup that holds more than 64 KiB of LDS does not
    survive being preempted for another process (7 of 88 backward passes corrupted with the 156-KiB halo kernels,
    0 of 60 with every kernel held to <= 64 KiB, 0 of 40 for a process that owns the GPU).  So when the ranks of this job do not
    each have a device of their own, every rank switches the library to its <= 64-KiB kernel variants and says so on stderr —
    slower (gather kernels instead of the halo-resident ones), never silently wrong.  Returns True when the guard engaged."""
    import sys
    mine = identity if identity is not None else _device_identity(local)
    everyone = [None] * dist.get_world_size(group)
    dist.all_gather_object(everyone, mine, group=group)
    shared = len(set(everyone)) < len(everyone)
    if shared:
        from . import _lib as L
        SHARED_DEVICE[0] = True
        if torch.cuda.is_available():
            lib = L.load()
            if lib.dm_set_conv_variant(2) != 0 or lib.dm_set_wgrad_variant(2) != 0:
                raise DmError(lib.dm_last_error().decode())
            L._guard["shared"] = True                 # (the per-process guard has nothing left to do)
        if dist.get_rank(group) == 0:
            print(f"[diffusionmodel_amd] {len(everyone)} ranks share {len(set(everyone))} device(s): selecting the <= 64-KiB-LDS "
                  "kernel variants (DM_CONV_VARIANT=2, DM_WGRAD_VARIANT=2); workgroups with more LDS are not preemption-safe "
                  "between processes on this stack. Use one process per GPU for full speed.", file=sys.stderr, flush=True)
    return shared


SHARED_DEVICE = [False]
_NOCOMM = os.environ.get("DM_DP_NOCOMM") == "1"


def bucket_bounds(total, n_buckets, align=1024):
    """Split [0, total) into <= n_buckets contiguous ranges with `align`-element boundaries."""
    n_buckets = max(1, min(n_buckets, max(1, total // align)))
    step = -(-total // n_buckets)
    step = -(-step // align) * align
    bounds, lo = [], 0
    while lo < total:
        hi = min(total, lo + step)
        bounds.append((lo, hi))
        lo = hi
    return bounds


class GradReducer:
    """Sum-all-reduce a flat gradient buffer in a few large buckets, last bucket first (the backward
    pass fills the buffer roughly back to front).  Works on CPU tensors with gloo (tests) and on
    device tensors with RCCL."""

    def __init__(self, flat_grad, n_buckets=4, group=None):
        self.flat = flat_grad
        self.group = group
        self.bounds = bucket_bounds(flat_grad.numel(), n_buckets)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = dist.is_initialized()
        self._works = []
        self._stream = torch.cuda.Stream() if flat_grad.is_cuda else None

    def start(self):
        """Launch the bucket all-reduces (asynchronously on a side stream for device tensors)."""
        if not self.active:
            return
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                for lo, hi in reversed(self.bounds):
                    self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            for lo, hi in reversed(self.bounds):
                self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Make the reduced gradients visible to the current stream."""
        for w in self._works:
            w.wait()
        self._works = []
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    def all_reduce(self):
        self.start()
        self.finish()


class OverlappedGradReducer:
    """Gradient all-reduce overlapped with the backward pass.

    The conv weight-gradient kernels write straight into the optimiser's flat gradient buffer (`param.main_grad`) and
    account for ~99 % of its bytes; the backward pass completes them back to front.  The buffer is cut into buckets on
    parameter boundaries; `ops.ON_WGRAD` tells the reducer when a parameter's weight gradient has been launched, and as
    soon as the last main-grad parameter of a bucket is in, the bucket's all-reduce starts on a side stream (ordered after
    the launch stream by an event) while the backward pass goes on.  The remaining parameters (biases, BatchNorm, dense
    layers: autograd `.grad` tensors that the optimiser folds in after backward) sit at zero in those early reductions;
    they are packed into one small contiguous buffer after backward, reduced once, and scattered back.

        red = OverlappedGradReducer(opt)          # after FusedAdamW, before the first step
        opt.zero_grad(); red.begin(); loss.backward(); red.finish(); opt.step()

    One all-reduce per bucket per step: no gradient accumulation over several backward passes in this mode.

    Launch plans (graph.GraphedTrainStep(mode="plan", body=..., runner=red.replay)): inside a stream capture `begin(capture=True)`
    makes the reducer emit dm_plan_marker launches where it would have started a collective — bucket b complete (marker b), backward
    pass over (MARK_BACKWARD_DONE), small gradients packed (MARK_SMALL_PACKED) — and `replay(plan)` re-issues the plan segment by
    segment from C, running the collectives between the segments exactly as the eager path orders them."""
    MARK_BACKWARD_DONE, MARK_SMALL_PACKED = 100000, 100001

    def __init__(self, opt, n_buckets=6, group=None):
        from . import ops
        self.opt, self.group, self._ops = opt, group, ops
        self.flat = opt.flat_g
        self.active = dist.is_initialized()
        # RCCL runs a collective on the process group's OWN stream (ordered after the issuing stream by an event, waited for by
        # Work.wait()): a second stream of ours in front of it is one more queue hand-off per bucket for nothing (DESIGN section 6:
        # the first cross-queue leg of a step costs ~0.3 ms whatever travels on it).  Other backends (gloo on device tensors: the
        # rehearsal tests) block the issuing stream, so they keep the side stream.  DM_DP_SIDE_STREAM=1/0 overrides.
        side = os.environ.get("DM_DP_SIDE_STREAM")
        use_side = (not self.active or dist.get_backend(group) != "nccl") if side is None else side != "0"
        self._stream = torch.cuda.Stream() if (self.flat.is_cuda and use_side) else None
        # RCCL orders a collective after the work already queued on the stream it is issued from; gloo stages device tensors
        # through the host on its own schedule, so with gloo (rehearsals on one GPU) the device is drained before every launch
        self._drain = self.active and self.flat.is_cuda and dist.get_backend(group) != "nccl"
        slots = opt._slots                                        # (param, offset, numel) in flat-buffer order
        main = [(i, p, off, n) for i, (p, off, n) in enumerate(slots) if hasattr(p, "main_grad")]
        total_main = sum(n for _, _, _, n in main)
        # bucket boundaries on parameter boundaries, ~equal main-grad bytes each
        self.buckets, lo_slot, acc = [], 0, 0
        target = max(1, total_main // max(1, n_buckets))
        for i, (p, off, n) in enumerate(slots):
            if hasattr(p, "main_grad"):
                acc += n
            last = i == len(slots) - 1
            if (acc >= target and len(self.buckets) < n_buckets - 1) or last:
                lo = slots[lo_slot][1]
                hi = off + (n + 3) // 4 * 4 if not last else self.flat.numel()
                self.buckets.append({"lo": lo, "hi": hi, "slots": range(lo_slot, i + 1)})
                lo_slot, acc = i + 1, 0
        # Which parameters report through ON_WGRAD (their weight-gradient kernel wrote the flat buffer) and which arrive as
        # autograd `.grad` tensors is a property of the layers' code paths, not of the parameter's shape (the 1x1 convolutions
        # of CoordAttn own a main_grad view but run as dense layers and deliver a `.grad`).  So the first step only observes:
        # every bucket is reduced after backward, the notifying parameters and the `.grad` parameters are recorded, and from
        # the second step on buckets launch as soon as their notifying parameters are in.
        self._slot_of = {id(p): i for i, (p, _, _) in enumerate(slots)}
        self._bucket_of_slot = [0] * len(slots)
        for b, bk in enumerate(self.buckets):
            for i in bk["slots"]:
                self._bucket_of_slot[i] = b
        self._learned, self._seen, self._seen_seq = False, set(), []
        self._bucket_of, self._need = {}, [0] * len(self.buckets)
        self._small, self._small_key, self._small_total = [], None, 0
        self._packed, self._tables = None, None
        self._pending, self._launched, self._works = [], [], []
        self._order, self._learned_order = [], None       # bucket launch order of this step / of the last complete overlapped step

    # ---- per step ----------------------------------------------------------------------------
    def begin(self, capture=False):
        self._capture = bool(capture)
        if self._capture and not self._learned:
            raise DmError("OverlappedGradReducer: run one eager step (the observation step) before capturing a plan")
        self._pending = list(self._need)
        self._launched = [False] * len(self.buckets)
        self._works = []
        self._seen, self._seen_seq = set(), []
        self._order = []
        self._ops.ON_WGRAD = self._notify if self.active else None

    def _notify(self, p):
        if not self._learned:                         # observation step: just record who reports, and in which order
            self._seen.add(id(p))
            self._seen_seq.append(id(p))
            return
        b = self._bucket_of.get(id(p))
        if b is None or self._launched[b]:
            return
        self._pending[b] -= 1
        if self._pending[b] <= 0:
            self._launch(b)

    def _launch(self, b):
        self._launched[b] = True
        self._order.append(b)
        if getattr(self, "_capture", False):           # recorded as a segment boundary; replay() runs the collective there
            from ._lib import call
            call("dm_plan_marker", int(b))
            return
        bk = self.buckets[b]
        view = self.flat[bk["lo"]:bk["hi"]]
        if self._drain:
            torch.cuda.synchronize()
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())          # the weight gradients launched so far
            if _NOCOMM:                                                    # (measurement aid: the choreography without the collectives)
                return
            with torch.cuda.stream(self._stream):
                self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            if _NOCOMM and self.flat.is_cuda:
                return
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _copy_small(self, to_packed):
        """flat <-> packed for the small parameters, one multi-tensor launch (CPU tensors: plain slicing)."""
        if not self.flat.is_cuda:
            o = 0
            for off, n in self._small:
                if to_packed:
                    self._packed[o:o + n] = self.flat[off:off + n]
                else:
                    self.flat[off:off + n] = self._packed[o:o + n]
                o += (n + 3) // 4 * 4
            return
        from ._lib import call, ptr
        if self._tables is None:
            rows_in, rows_out, o = [], [], 0
            fb, pb = self.flat.data_ptr(), self._packed.data_ptr()
            for off, n in self._small:
                for lo in range(0, n, 16384):
                    m = min(16384, n - lo)
                    rows_in.append((fb + 4 * (off + lo), pb + 4 * (o + lo), m))
                    rows_out.append((pb + 4 * (o + lo), fb + 4 * (off + lo), m))
                o += (n + 3) // 4 * 4
            dev = self.flat.device
            self._tables = (torch.tensor(rows_in, dtype=torch.int64).to(dev), torch.tensor(rows_out, dtype=torch.int64).to(dev), len(rows_in))
        tin, tout, n_rows = self._tables
        call("dm_scatter_copy", ptr(tin if to_packed else tout), n_rows, 0)

    def finish(self, idle=False):
        """Call after backward: reduces what is left (buckets nobody completed, the small parameters) and makes the summed
        gradients visible to the launch stream.  Folds the autograd `.grad` tensors into the flat buffer on the way.
        idle=True: this rank ran NO backward pass under the reducer for this step (short tail of an epoch: its peers did) — it
        issues the same collectives in the order an overlapped step issues them, contributing whatever its buffers hold."""
        self._ops.ON_WGRAD = None
        settle = getattr(self.opt, "settle_fresh", None)
        if settle is not None:
            settle()                                  # lazily-zeroed gradient ranges nobody wrote (an idle rank: all of them) are zeros now
        if not self.active:
            return
        if idle:
            if getattr(self, "_capture", False) or not self._learned or self._small_key is None:
                raise DmError("OverlappedGradReducer.finish(idle=True): needs one completed eager step on this rank first "
                              "(the first optimiser step of a run cannot be a short tail)")
            for b in (self._learned_order or range(len(self.buckets))):
                if not self._launched[b]:
                    self._launch(b)
        if getattr(self, "_capture", False):
            from ._lib import call
            call("dm_plan_marker", self.MARK_BACKWARD_DONE)
            small = tuple((off, n) for (p, off, n) in self.opt._slots if p.grad is not None)
            if small != self._small_key:
                raise DmError("OverlappedGradReducer: the set of `.grad` parameters changed between the observation step and the capture")
            self.opt.gather_grads()
            if self._small_total:
                self._copy_small(True)
                call("dm_plan_marker", self.MARK_SMALL_PACKED)
                self._copy_small(False)
            return
        if not self._learned:
            self._bucket_of = {k: self._bucket_of_slot[self._slot_of[k]] for k in self._seen if k in self._slot_of}
            self._need = [0] * len(self.buckets)
            for b in self._bucket_of.values():
                self._need[b] += 1
            self._need = [n if n > 0 else 1 << 30 for n in self._need]     # a bucket nobody reports into waits for finish()
            self._learned = True
            # the order in which an overlapped step will launch the buckets (notification order, then the rest by index): what a
            # rank without a backward pass of its own (finish(idle=True)) must follow so that the collectives pair up
            pend, order = list(self._need), []
            for k in self._seen_seq:
                b = self._bucket_of.get(k)
                if b is not None and b not in order:
                    pend[b] -= 1
                    if pend[b] <= 0:
                        order.append(b)
            self._learned_order = order + [b for b in range(len(self.buckets)) if b not in order]
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
        if not idle and len(self._order) == len(self.buckets):
            self._learned_order = list(self._order)
        # the early reductions summed zeros in the `.grad` slots; nothing else may touch those slots until they are done
        for w in self._works:
            w.wait()
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
        self._works = []
        # the parameters whose gradient is an autograd `.grad` right now are exactly the slots the buckets have not covered
        small = [(off, n) for (p, off, n) in self.opt._slots if p.grad is not None]
        if idle and not small:
            small = list(self._small)                 # nothing of its own to add: the learned set, so that the packed sizes agree
        key = tuple(small)
        if key != self._small_key:
            self._small, self._small_key, self._tables = small, key, None
            self._small_total = sum((n + 3) // 4 * 4 for _, n in small)
            self._packed = torch.zeros(max(self._small_total, 4), dtype=torch.float32, device=self.flat.device)
        self.opt.gather_grads()                       # local `.grad` gradients into their (still local) slots
        if self._small_total:
            self._copy_small(True)
            if self._drain:
                torch.cuda.synchronize()
            dist.all_reduce(self._packed, op=dist.ReduceOp.SUM, group=self.group)
            self._copy_small(False)

    def all_reduce(self):                             # drop-in for GradReducer when nothing was overlapped
        self.finish()

    def replay(self, plan, run=None):
        """One planned step: the segments of `plan` (graph.LaunchPlan) from C, the collectives of the eager path between them.
        `run(first, last)` replaces plan.run (bench.py: the timed variant)."""
        self._capture = False
        self._launched = [False] * len(self.buckets)
        self._works = []
        run = plan.run if run is None else run
        for s in range(plan.n_segments):
            run(s, s)
            m = plan.segment_markers[s]
            if 0 <= m < len(self.buckets):
                self._launch(m)
            elif m == self.MARK_BACKWARD_DONE:
                timing = getattr(self, "time_tail", False)
                if timing:                            # launch-stream time between the last backward kernel and the reduced buckets
                    self._tail_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    self._tail_ev[0].record()
                for b in range(len(self.buckets)):
                    if not self._launched[b]:
                        self._launch(b)
                for w in self._works:
                    w.wait()
                if self._stream is not None:
                    torch.cuda.current_stream().wait_stream(self._stream)
                if timing:
                    self._tail_ev[1].record()
                self._works = []
            elif m == self.MARK_SMALL_PACKED:
                if self._drain:
                    torch.cuda.synchronize()
                if not _NOCOMM:
                    dist.all_reduce(self._packed, op=dist.ReduceOp.SUM, group=self.group)
            elif m != -1:
                raise DmError(f"OverlappedGradReducer.replay: unknown plan marker {m}")


    def tail_wait_ms(self):
        """With `time_tail = True`: launch-stream milliseconds of the last replay between the end of the backward pass and the
        moment every bucket's all-reduce had finished (what the overlap did NOT hide).  Synchronises."""
        ev = getattr(self, "_tail_ev", None)
        if ev is None:
            return None
        ev[1].synchronize()
        return float(ev[0].elapsed_time(ev[1]))

    def bytes_per_step(self):
        """Bytes this rank hands to all-reduce per optimiser step: the flat fp32 gradient in buckets + the packed small gradients."""
        return 4 * (int(self.flat.numel()) + int(self._small_total))


class CAbiComm:
    """The C ABI's own collective (include/dm_amd.h: dm_comm_unique_id / dm_comm_init / dm_allreduce_bucket / dm_comm_destroy —
    RCCL bound by dlopen inside libdm_amd.so), for hosts that bring no torch.distributed.  `exchange(id_bytes_or_None) -> id_bytes`
    is the caller's out-of-band channel for the 128-byte unique id (rank 0 passes its id in, every rank gets rank 0's back); with
    world == 1 it is not needed.  all_reduce() is an in-place SUM on torch's current stream, asynchronous like a kernel launch."""

    def __init__(self, rank=0, world=1, exchange=None):
        import ctypes as C
        from . import _lib as L
        lib = L.load()
        uid = C.create_string_buffer(128)
        if rank == 0 and lib.dm_comm_unique_id(uid) != 0:
            raise DmError("dm_comm_unique_id: " + lib.dm_last_error().decode())
        if world > 1:
            if exchange is None:
                raise DmError("CAbiComm: world > 1 needs an `exchange` callable that carries rank 0's 128-byte id to every rank")
            uid = C.create_string_buffer(bytes(exchange(uid.raw if rank == 0 else None)), 128)
        self._h, self._lib, self.rank, self.world = C.c_void_p(), lib, rank, world
        if lib.dm_comm_init(C.byref(self._h), int(world), int(rank), uid) != 0:
            raise DmError("dm_comm_init: " + lib.dm_last_error().decode())

    def all_reduce(self, t):
        from . import ops
        if not t.is_contiguous():
            raise DmError("CAbiComm.all_reduce: contiguous tensors only")
        ops.call("dm_allreduce_bucket", ops.ptr(t), t.numel(), ops.dt(t), self._h)
        return t

    def close(self):
        if self._h:
            self._lib.dm_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:                         # noqa: BLE001 — interpreter shutdown
            pass


def wait_ranks(procs, poll=0.2):
    """Wait for the child ranks; when one exits non-zero the others are terminated (they would otherwise sit in a collective
    until the RCCL timeout).  Returns the list of exit codes."""
    import time
    codes = [None] * len(procs)
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(timeout=20)
                    except Exception:                 # noqa: BLE001
                        p.kill()
                        codes[i] = p.wait()
            break
        time.sleep(poll)
    return codes


def launch_ranks(script, argv, n, extra_env=None):
    """Start `n` fresh child processes of `script` — one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set — and return
    the first non-zero exit code (0 if none).  The CALLER must not have made a GPU call (torch.cuda.device_count() is fine on this
    image): it only waits.  Rank 0 inherits stdout, the other ranks' stdout goes to stderr."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    codes = wait_ranks(procs)
    bad = [c for c in codes if c]
    return bad[0] if bad else 0


def broadcast_buffers(module, src=0, group=None):
    """Rank `src`'s floating-point buffers (BatchNorm running statistics: every rank keeps the statistics of ITS micro-batches,
    SURVEY 8e) to every rank — before a validation / sampling pass whose result must not depend on the rank."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    bufs = [b for b in module.buffers() if b.is_floating_point()]
    if not bufs:
        return
    flat = torch.cat([b.detach().reshape(-1).float() for b in bufs])
    if flat.is_cuda and dist.get_backend(group) != "nccl":
        host = flat.cpu()
        dist.broadcast(host, src=src, group=group)
        flat = host.to(flat.device)
    else:
        dist.broadcast(flat, src=src, group=group)
    o = 0
    with torch.no_grad():
        for b in bufs:
            b.copy_(flat[o:o + b.numel()].view_as(b))
            o += b.numel()
    for m in module.modules():                  # eval-mode folds are cached per statistics epoch
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m._stat_epoch = getattr(m, "_stat_epoch", 0) + 1


def broadcast_parameters(flat_params, src=0, group=None):
    """Identical initial weights on every rank."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)


def sample_sharded(ddpm, n_sample, size, device, guide_w=0.0, *, group=None, gather=True, **kw):
    """CFG sampling sharded over the ranks (samples are independent, no exchange inside the trajectory).

    Rank r runs `ddpm.sample` on samples [r m, (r + 1) m), m = n_sample / world, of the class-cycled batch with their slice of
    the Philox noise stream; one all-gather at the end returns the (n_sample, *size) images on every rank (gather=False: the
    local shard only).  Same seed on every rank -> the result does not depend on the world size (eval-mode BatchNorm is
    per-sample; only the split-K summation order of a few layers follows the batch size)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if n_sample % world:
        raise DmError(f"n_sample={n_sample} does not shard over {world} ranks")
    if kw.get("seed") is None:
        raise DmError("sample_sharded needs an explicit seed= (identical on every rank)")
    m = n_sample // world
    x = ddpm.sample(m, size, device, guide_w, first_sample=rank * m, total_samples=n_sample, **kw)
    if world == 1 or not gather:
        return x
    out = torch.empty((n_sample,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out

