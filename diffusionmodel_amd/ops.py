"""Operator layer: torch.autograd.Function wrappers over the C-ABI kernels (include/dm_amd.h).

Conventions
-----------
* activations are NHWC tensors `(B, H, W, C)`, contiguous, dtype float32 ("fp32 mode", exact-fp32 MFMA,
  the 1e-4 parity mode), bfloat16 ("bf16 mode", the throughput mode) or float16 (the reference's autocast dtype; needs the
  dynamic loss scaling of optim.DmGradScaler); C % 8 == 0;
* parameters stay fp32 in the reference's shapes (OIHW ...) but in channels_last memory format, i.e.
  physically [O][kh][kw][I] — exactly the K-contiguous row layout the implicit-GEMM kernel reads,
  so fp32 mode uses the master weights in place and bf16 mode needs only an elementwise cast;
* strips / gates / embeddings / statistics are fp32;
* every kernel goes to torch's current stream through ctypes; torch itself is used for allocation
  and autograd bookkeeping only.
"""
import ctypes as C
import threading

import weakref

import torch

from . import _lib as L
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, call, dt, ptr

BN_EPS = 1e-5
import os as _os
FUSED_CHAINS = _os.environ.get("DM_FUSED_CHAINS", "1") != "0"    # SE MLP / CoordAttn strip chains as 1-2 launches per direction (chain.hip); 0: one launch per torch.nn op (A/B measurements)
BN_SLOTS_MAX_C = int(_os.environ.get("DM_BN_SLOTS_MAX_C", "256"))   # wider layers keep the partial rows + finalize launch: every workgroup of the consuming kernel folds slots x C doubles (128 KiB at C = 1024: 8x the tensor's own traffic on the 8x8 level, +6 us per launch against 2.6 us for the finalize launch)
BN_SLOTS = int(_os.environ.get("DM_BN_SLOTS", "8"))          # train-mode BatchNorm statistics travel as [BN_SLOTS][C] accumulators folded by the consuming kernel (0: partial rows + finalize launches)
_EPOCH = [0]          # bumped by the fused optimiser: invalidates packed-weight caches
_CACHE = {}
ON_WGRAD = None       # parallel.GradReducer: called with the parameter whose main_grad a weight-gradient launch just completed
PROFILE_KINDS = ("conv_igemm", "conv_wgrad", "igemm_f32", "wgrad_f32")   # bench.py narrows this to the kinds it reports
PROFILE = None        # bench.py sets this to a list to collect (kind, flops, start_event, end_event, shape)
PROFILE_META = None   # graph.GraphedTrainStep: (kind, flops, shape) per MFMA launch of the captured step, no events


def bump_weight_epoch():
    _EPOCH[0] += 1
    _CACHE.clear()


def _empty(shape, dtype, ref):
    return torch.empty(shape, dtype=dtype, device=ref.device)


def _zeros(shape, dtype, ref):
    return torch.zeros(shape, dtype=dtype, device=ref.device)


class _ZeroArena:
    """Pre-zeroed fp32 scratch for the many small gradient accumulators of a backward pass (bias / BatchNorm /
    dense-layer gradients that the kernels add into).  One fill per optimiser step instead of one per tensor: slices
    are handed out in order and FusedAdamW.step() — which has by then copied every gradient into its flat buffer —
    re-zeroes the used part.  Without that optimiser (or once the arena is full) `take` falls back to torch.zeros."""
    CAP = 48 << 20      # bytes

    def __init__(self):
        self.buf, self.off = None, 0

    def take(self, shape, ref):
        n = 1
        for d in shape:
            n *= int(d)
        if self.buf is None or self.buf.device != ref.device:
            if self.buf is not None:
                return torch.zeros(shape, dtype=torch.float32, device=ref.device)
            self.buf, self.off = torch.zeros(self.CAP // 4, dtype=torch.float32, device=ref.device), 0
        lo = (self.off + 3) // 4 * 4                  # 16-byte aligned slices
        if lo + n > self.buf.numel():
            return torch.zeros(shape, dtype=torch.float32, device=ref.device)
        # (also inside a stream capture: the arena is a static buffer, the slices are handed out in the same order on every
        #  replay and recycle()'s one fill is captured with them — the fallback would put ~130 fill launches into every step)
        self.off = lo + n
        return self.buf[lo:lo + n].view(shape)

    def recycle(self):
        """Everything handed out so far is dead (its values were consumed): zero it again and start over."""
        if self.buf is not None and self.off:
            self.buf[:self.off].zero_()
        self.off = 0


ZERO_ARENA = _ZeroArena()
ARENA_ENABLED = [False]     # switched on by FusedAdamW (the only component that knows when the gradients are dead)


def _gzeros(shape, ref):
    """fp32 zeros for a gradient accumulator that is consumed before the next optimiser step."""
    if ARENA_ENABLED[0]:
        return ZERO_ARENA.take(shape, ref)
    return torch.zeros(shape, dtype=torch.float32, device=ref.device)


class GradFork:
    """A tensor with TWO consumers (the skip / residual / attention forks of new_scripy.py:196-205, 242, 320-355).  autograd would
    add the two gradients with an elementwise kernel of its own; instead the LATER consumer (which runs its backward FIRST — in
    every fork of this network it also consumes something computed from the earlier consumer's output) leaves its gradient here
    (`second(x)`), and the earlier consumer's input-gradient kernel adds it in its epilogue (`take()` -> DmConv.addend / the
    `dout_gate` operand of dm_ca_pool_bwd / dm_add)."""
    __slots__ = ("stash",)

    def __init__(self):
        self.stash = None

    def second(self, x):
        """The later consumer reads x through this: identity forward, the gradient is stashed instead of returned."""
        self.stash = None
        if torch.is_grad_enabled() and x.requires_grad:
            return _StashGrad.apply(x, self)
        return x

    def take(self):
        g, self.stash = self.stash, None
        return g


class _StashGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fork):
        ctx.fork = fork
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if ctx.fork.stash is not None:
            # the earlier consumer's backward never took the previous stash: a second backward over the same graph
            # (retain_graph=True), or autograd.grad on a sub-graph that excludes the earlier consumer — the gradient would be lost
            raise L.DmError("GradFork: the stashed gradient of a forked tensor was never consumed (a second backward pass over the "
                            "same graph, or a sub-graph without the tensor's first consumer, is not supported on the HIP path)")
        ctx.fork.stash = g.contiguous()
        return None, None


def _cl(w):
    """Physical [O][kh][kw][I] view of a 4-D parameter (rewrites its storage once if needed)."""
    if w.dim() == 4 and not w.is_contiguous(memory_format=torch.channels_last):
        with torch.no_grad():
            w.data = w.data.contiguous(memory_format=torch.channels_last)
    return w


def _cached(key, w, build):
    """Derived-tensor cache that lives ON the parameter object (so it dies with it; a data_ptr-keyed
    global cache would hand a new tensor at a recycled address somebody else's pack)."""
    stamp = (w.data_ptr(), w._version, _EPOCH[0])
    store = getattr(w, "_dm_cache", None)
    if store is None or store[0] != stamp:
        store = (stamp, {})
        w._dm_cache = store
    hit = store[1].get(key)
    if hit is None:
        hit = build()
        store[1][key] = hit
    return hit


def packed_fwd(w, dtype, cp):
    """[N][T][cp] weights in `dtype` for dm_conv. fp32 without padding is the master itself; bf16 without padding is
    the optimiser's bf16 shadow of the parameter when there is one (FusedAdamW refreshes it inside its own kernel)."""
    _cl(w)
    n, c = w.shape[0], w.shape[1]
    t = w.shape[2] * w.shape[3]
    if dtype == torch.float32 and cp == c:
        return w
    shadow = getattr(w, "_dm_shadow16", None)
    if shadow is not None and dtype == shadow.dtype and cp == c:
        stamp = (w.data_ptr(), w._version)
        if w._dm_shadow_stamp != stamp:          # modified behind the optimiser's back (load_state_dict, manual init ...)
            call("dm_cast", ptr(w), ptr(shadow), L.DM_F32, dt(shadow), w.numel())
            w._dm_shadow_stamp = stamp
        return shadow

    def build():
        out = _empty((n, t, cp), dtype, w)
        call("dm_pack_w", ptr(w), ptr(out), dt(dtype), n, t, c, cp)
        return out
    return _cached(("fwd", dtype, cp), w, build)


def packed_T(w, dtype, taps, np_):
    """[C][len(taps)][np_] transposed weights for the input-gradient GEMM."""
    _cl(w)
    n, c = w.shape[0], w.shape[1]
    t = w.shape[2] * w.shape[3]
    taps = list(range(t)) if taps is None else list(taps)
    return _packed_T_core(w, ("T", dtype, tuple(taps), np_), n, t, c, taps, np_, dtype)


def _packed_T_core(w, key, n, t, c, taps, np_, dtype):
    def build():
        reg = _T_PACKS.get((id(w), key))
        if reg is not None and reg[0]() is w:
            out = reg[1]                          # re-pack in place: the batched refresh keeps pointing at this tensor
        else:
            out = _empty((c, len(taps), np_), dtype, w)
            if len(taps) <= 16 and isinstance(w, torch.nn.Parameter):      # (views / temporaries stay on the lazy path)
                _T_PACKS[(id(w), key)] = (weakref.ref(w), out, (n, t, c, tuple(taps), np_, dt(dtype)))
                _T_TABLES.clear()
        arr = (C.c_int32 * len(taps))(*taps)
        call("dm_pack_wT", ptr(w), ptr(out), dt(dtype), n, t, c, len(taps), arr, np_)
        return out
    return _cached(key, w, build)


_T_PACKS = {}        # (id(param), key) -> (weakref(param), packed tensor, geometry): every transposed pack ever requested
_T_TABLES = {}       # device tables of dm_pack_multi, rebuilt when the registry changes


def refresh_packs():
    """Re-pack every registered transposed weight pack with ONE launch (after an optimiser step) and re-stamp the
    per-parameter caches, so the backward pass of the next step finds them fresh."""
    if not _T_PACKS:
        return
    if not _T_TABLES:
        ents, taps_rows, blocks, live_keys, live_ptrs = [], [], [], [], []
        for k in list(_T_PACKS):
            wr, out, (n, t, c, taps, np_, dty) = _T_PACKS[k]
            w = wr()
            if w is None:
                del _T_PACKS[k]
                continue
            e = len(ents)
            live_keys.append(k)
            live_ptrs.append(w.data_ptr())
            sh = getattr(w, "_dm_shadow16", None)         # bf16 copy of the parameter kept current by the fused optimiser step
            if sh is not None and dty == dt(sh):          # 0x100 / 0x200: the source is the bf16 / fp16 shadow
                ents.append((sh.data_ptr(), out.data_ptr(), n, t, c, len(taps), np_, dty | (0x100 if dty == L.DM_BF16 else 0x200)))
            else:
                ents.append((w.data_ptr(), out.data_ptr(), n, t, c, len(taps), np_, dty))
            taps_rows.append(list(taps) + [0] * (16 - len(taps)))
            for tt in range(len(taps)):
                for ct in range((c + 63) // 64):
                    for nt in range((np_ + 63) // 64):
                        blocks.append((e, nt, ct, tt))
        if not ents:
            return
        dev = next(iter(_T_PACKS.values()))[1].device
        _T_TABLES["ents"] = torch.tensor(ents, dtype=torch.int64).to(dev)
        _T_TABLES["taps"] = torch.tensor(taps_rows, dtype=torch.int32).to(dev)
        _T_TABLES["blocks"] = torch.tensor(blocks, dtype=torch.int32).to(dev)
        _T_TABLES["ptrs"] = live_ptrs                     # (collected while the parameters were known to be alive: a garbage
        _T_TABLES["keys"] = live_keys                     #  collection during the uploads above may retire another model's)
    # a parameter whose storage moved since the tables were built falls back to the lazy per-tensor path
    keys = _T_TABLES["keys"]
    for k, p0 in zip(keys, _T_TABLES["ptrs"]):
        w = _T_PACKS[k][0]() if k in _T_PACKS else None
        if w is None or w.data_ptr() != p0:
            _T_TABLES.clear()
            return
    call("dm_pack_multi", ptr(_T_TABLES["ents"]), ptr(_T_TABLES["taps"]), ptr(_T_TABLES["blocks"]), _T_TABLES["blocks"].shape[0])
    for k in keys:
        wr, out, _ = _T_PACKS[k]
        w = wr()
        stamp = (w.data_ptr(), w._version, _EPOCH[0])
        store = getattr(w, "_dm_cache", None)
        if store is None or store[0] != stamp:
            store = (stamp, {})
            w._dm_cache = store
        store[1][k[1]] = out


# ------------------------------------------------------------------------------------------------
# dm_conv / dm_conv_wgrad descriptors
# ------------------------------------------------------------------------------------------------
_DESC_TLS = threading.local()


def _desc_cache():
    c = getattr(_DESC_TLS, "cache", None)
    if c is None:
        c = _DESC_TLS.cache = {}
    return c


def _conv_desc(in1, in2, w_ptr, ldw, out, *, dtype, B, Hi, Wi, C1, C2, Hq, Wq, sy, sx, T, KW, ty, tx, oy0, ox0,
               Ho, Wo, N, osy=1, osx=1, ooy=0, oox=0, ldc=None, coff=0, scale=None, shift=None, act=ACT_NONE,
               psum=None, psq=None, out_nchw=False, in2_batch=0, addend=None, stat_slots=0):
    """The filled DmConv descriptor of one launch.  The geometry half is the same every step: one ctypes struct per distinct call,
    only the pointers change (filling 37 fields costs ~6 us of Python per launch; the backward pass runs on autograd's thread, hence
    a cache per thread)."""
    key = (dtype, B, Hi, Wi, C1, C2, Hq, Wq, sy, sx, T, KW, ty, tx, oy0, ox0, Ho, Wo, N, osy, osx, ooy, oox, ldc, coff, act,
           out_nchw, in2_batch, ldw, stat_slots)
    cache = _desc_cache()
    d = cache.get(key)
    if d is None:
        d = L.DmConv()
        d.dtype, d.act, d.out_nchw_f32 = dt(dtype), act, int(out_nchw)
        d.B, d.Hi, d.Wi, d.C1, d.C2 = B, Hi, Wi, C1, C2
        d.Hq, d.Wq, d.sy, d.sx = Hq, Wq, sy, sx
        d.T, d.KW, d.ty, d.tx, d.oy0, d.ox0 = T, KW, ty, tx, oy0, ox0
        d.Ho, d.Wo, d.osy, d.osx, d.ooy, d.oox = Ho, Wo, osy, osx, ooy, oox
        d.N, d.ldw, d.ldc, d.coff = N, ldw, (N if ldc is None else ldc), coff
        d.in2_batch = in2_batch
        d.stat_slots = stat_slots
        cache[key] = d
    d.in1, d.in2, d.w = ptr(in1), ptr(in2), w_ptr
    d.scale, d.shift, d.out, d.psum, d.psq = ptr(scale), ptr(shift), ptr(out), ptr(psum), ptr(psq)
    d.addend = ptr(addend)
    return d


def _conv_call(in1, in2, w_ptr, ldw, out, **kw):
    L.ensure_workspace()          # split-K partial tiles of small, deep problems
    d = _conv_desc(in1, in2, w_ptr, ldw, out, **kw)
    dtype, B, Hi, Wi, C1, C2, Hq, Wq, N, T, sy, ty = (kw[k] for k in ("dtype", "B", "Hi", "Wi", "C1", "C2", "Hq", "Wq", "N", "T", "sy", "ty"))
    kind = "conv_igemm" if dtype != torch.float32 else "igemm_f32"
    if PROFILE_META is not None:
        call("dm_conv", C.byref(d))
        halo = kind == "conv_igemm" and L.load().dm_last_conv_path() == 1
        PROFILE_META.append(("conv_halo" if halo else kind, 2.0 * B * Hq * Wq * N * T * (C1 + C2), f"B{B} {Hi}x{Wi} C{C1}+{C2} N{N} T{T} s{sy} t{ty}"))
    elif PROFILE is None or kind not in PROFILE_KINDS:
        call("dm_conv", C.byref(d))
    else:   # bench.py: HIP events on the launch stream around this launch + its algorithmic FLOPs
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("dm_conv", C.byref(d))
        e1.record()
        if kind == "conv_igemm" and L.load().dm_last_conv_path() == 1:
            kind = "conv_halo"                # the launch went to conv3x3_halo_kernel
        PROFILE.append((kind, 2.0 * B * Hq * Wq * N * T * (C1 + C2), e0, e1,
                        f"B{B} {Hi}x{Wi} C{C1}+{C2} N{N} T{T} s{sy} t{ty} o{kw['oy0']} out{kw['Ho']}x{kw['Wo']}/{kw.get('osy', 1)}"))


def _conv_parity4_call(calls):
    """Four (in1, in2, w_ptr, ldw, out, kwargs) launches that differ only in weights / tap offsets / output offsets — the four
    output-parity classes of a stride-2 layer's input gradient: one dm_conv_parity4 call (one launch where the four-tap halo kernel
    takes it, four otherwise)."""
    L.ensure_workspace()
    arr = (L.DmConv * 4)()
    flops = 0.0
    for i, (in1, in2, w_ptr, ldw, out, kw) in enumerate(calls):
        d = _conv_desc(in1, in2, w_ptr, ldw, out, **kw)
        C.memmove(C.byref(arr, i * C.sizeof(L.DmConv)), C.byref(d), C.sizeof(L.DmConv))
        flops += 2.0 * kw["B"] * kw["Hq"] * kw["Wq"] * kw["N"] * kw["T"] * (kw["C1"] + kw["C2"])
    kw = calls[0][5]
    kind = "conv_igemm" if kw["dtype"] != torch.float32 else "igemm_f32"
    if PROFILE is not None and kind in PROFILE_KINDS:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("dm_conv_parity4", arr)
        e1.record()
        PROFILE.append((kind, flops, e0, e1, f"B{kw['B']} {kw['Hi']}x{kw['Wi']} C{kw['C1']} N{kw['N']} T4 x4 parity classes"))
    else:
        call("dm_conv_parity4", arr)
        if PROFILE_META is not None:
            PROFILE_META.append((kind, flops, f"B{kw['B']} {kw['Hi']}x{kw['Wi']} C{kw['C1']} N{kw['N']} T4 x4 parity classes"))


def _wgrad_call(dy, in1, in2, dw, dbias, *, dtype, B, Hi, Wi, C1, C2, Hq, Wq, sy, sx, T, KW, ty, tx, oy0, ox0, Ho, Wo, N, ldy,
                ldw, osy=1, osx=1, ooy=0, oox=0, splitk=0, overwrite=False):
    L.ensure_workspace()          # split partial sums of the halo-resident weight-gradient kernel
    key = ("w", dtype, B, Hi, Wi, C1, C2, Hq, Wq, sy, sx, T, KW, ty, tx, oy0, ox0, Ho, Wo, N, ldy, ldw, osy, osx, ooy, oox, splitk)
    cache = _desc_cache()
    d = cache.get(key)
    if d is None:
        d = L.DmWgrad()
        d.dtype = dt(dtype)
        d.B, d.Hi, d.Wi, d.C1, d.C2 = B, Hi, Wi, C1, C2
        d.Hq, d.Wq, d.sy, d.sx = Hq, Wq, sy, sx
        d.T, d.KW, d.ty, d.tx, d.oy0, d.ox0 = T, KW, ty, tx, oy0, ox0
        d.Ho, d.Wo, d.osy, d.osx, d.ooy, d.oox = Ho, Wo, osy, osx, ooy, oox
        d.N, d.ldy, d.ldw, d.splitk = N, ldy, ldw, splitk
        cache[key] = d
    d.dy, d.in1, d.in2, d.dw, d.dbias = ptr(dy), ptr(in1), ptr(in2), ptr(dw), ptr(dbias)
    d.overwrite = 1 if overwrite else 0          # dw holds garbage (FusedAdamW's lazy zero_grad): the launch leaves dw = this gradient
    kind = "conv_wgrad" if dtype != torch.float32 else "wgrad_f32"
    if PROFILE_META is not None:
        call("dm_conv_wgrad", C.byref(d))
        halo = kind == "conv_wgrad" and L.load().dm_last_wgrad_path() in (1, 3)      # 3: the four-tap form of the same kernel (4x4 / stride 2)
        PROFILE_META.append(("wgrad_halo" if halo else kind, 2.0 * B * Hq * Wq * N * T * (C1 + C2), f"B{B} {Hi}x{Wi} C{C1}+{C2} N{N} T{T} s{sy}"))
    elif PROFILE is None or kind not in PROFILE_KINDS:
        call("dm_conv_wgrad", C.byref(d))
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("dm_conv_wgrad", C.byref(d))
        e1.record()
        PROFILE.append((kind, 2.0 * B * Hq * Wq * N * T * (C1 + C2), e0, e1,
                        f"B{B} {Hi}x{Wi} C{C1}+{C2} N{N} T{T} s{sy} q{Hq}x{Wq}"))


class ConvSpec:
    """Static description of one conv (+ optional BatchNorm + activation) site."""

    def __init__(self, kh, kw, stride, pad, act=ACT_NONE, bn=None, out_nchw=False):
        self.kh, self.kw, self.stride, self.pad, self.act, self.bn, self.out_nchw = kh, kw, stride, pad, act, bn, out_nchw
        self.nbt_pending = 0   # BatchNorm num_batches_tracked increments not yet flushed to the buffer


def _bn_eval_fold(spec, bias):
    bn = spec.bn
    key = ("fold", bias.data_ptr() if bias is not None else 0, bn.weight._version, bn.bias._version,
           bn.running_mean._version, bn.running_var._version, getattr(bn, "_stat_epoch", 0))

    def build():
        n = bn.weight.shape[0]
        sc, sh = _empty((n,), torch.float32, bn.weight), _empty((n,), torch.float32, bn.weight)
        call("dm_bn_fold", ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var), ptr(bias), BN_EPS, n,
             ptr(sc), ptr(sh))
        return sc, sh
    return _cached(key, bn.weight, build)


def folded_dense_bn(conv, spec):
    """Eval-mode BatchNorm folded INTO a 1x1-conv / dense layer's weights: (W', b') with BN(W x + b) = W' x + b' — inference only,
    rebuilt when the weights or the statistics change (CoordAttn's conv1_* + bn1_* on strips: one launch instead of two)."""
    sc, sh = _bn_eval_fold(spec, conv.bias)
    w = conv.weight
    key = ("dense_fold", w._version, sc.data_ptr())

    def build():
        with torch.no_grad():
            return (w.reshape(w.shape[0], -1).float() * sc[:, None]).contiguous()
    return _cached(key, w, build), sh


def _running_stats(bn, n):
    """(mean, rstd) of an eval-mode BatchNorm from its running statistics (dm_bn_fold with gamma=1, beta=0)."""
    key = ("rstats", bn.running_mean._version, bn.running_var._version, getattr(bn, "_stat_epoch", 0))

    def build():                                  # once per set of statistics: the sampler calls this every step
        ones, zeros = torch.ones_like(bn.running_var), torch.zeros_like(bn.running_var)
        rstd, dummy = torch.empty_like(bn.running_var), torch.empty_like(bn.running_var)
        call("dm_bn_fold", ptr(ones), ptr(zeros), ptr(zeros), ptr(bn.running_var), None, BN_EPS, n, ptr(rstd), ptr(dummy))
        return bn.running_mean.clone(), rstd
    return _cached(key, bn.running_var, build)


def conv_bcast_ok(dtype, B, H, W, C1, C2):
    """Can a 3x3 stride-1 conv read its second source with a smaller batch (DmConv.in2_batch)?  Mirrors halo_eligible() in
    igemm.hip: bf16, whole 64-channel chunks, rows of 16 / 32 / 64 pixels (or multiples of 64), four 8x8 images per tile."""
    if dtype == torch.float32 or C1 % 64 or C2 % 64 or L.load().dm_get_conv_variant() < 5:
        return False
    if W == 8:
        return H == 8 and B % 4 == 0
    if W > 64:
        return W % 64 == 0 and H % 4 == 0
    return W in (16, 32, 64) and (H * W) % 256 == 0


class ConvBnAct(torch.autograd.Function):
    """y = act(BN(conv(cat(x, x2)) + b)) as ONE autograd node.

    train + BN : dm_conv (raw z, per-block column statistics in the epilogue) -> dm_bn_finalize ->
                 dm_bn_act_fwd; saves x, z and the batch statistics.
    eval  + BN : BN folded into the conv epilogue (scale/shift/act) when no gradient is needed.
    no BN      : bias (+act) in the epilogue.
    backward   : BN/act backward (two streaming passes), input gradient as a gather-GEMM of dz with
                 transposed packed weights (per output-parity class for strided convs), weight gradient
                 by dm_conv_wgrad (fp32 atomics; accumulates straight into `w.main_grad` when present).
    """

    @staticmethod
    def forward(ctx, x, x2, w, b, gamma, beta, spec, need_grad=True, fork=None):
        L.require_device(x, w)
        ctx.fork = fork                                        # GradFork of x: another consumer's gradient of x is added in the dgrad epilogue
        x = x.contiguous()
        x2 = x2.contiguous() if x2 is not None else None
        B, Hi, Wi, C1 = x.shape
        C2 = x2.shape[3] if x2 is not None else 0
        N, Cin = w.shape[0], w.shape[1]
        dtype = x.dtype
        bcast = x2 is not None and x2.shape[0] != B           # CFG sampler: skip tensor of n samples under a batch of 2n
        if bcast and (need_grad or B % x2.shape[0] or spec.bn is not None and spec.bn.training):
            raise L.DmError(f"conv: second source with batch {x2.shape[0]} under batch {B} is inference-only (and must divide it)")
        s, p = spec.stride, spec.pad
        Ho = (Hi + 2 * p - spec.kh) // s + 1
        Wo = (Wi + 2 * p - spec.kw) // s + 1
        T = spec.kh * spec.kw
        Cp = C1 + C2
        if not (Cp == Cin or (C2 == 0 and Cp > Cin and Cp - Cin < 8)):
            raise L.DmError(f"conv: input has {Cp} channels, weight expects {Cin}")
        if not spec.out_nchw and N % 8:
            raise L.DmError(f"conv: Cout={N} must be a multiple of 8 on the HIP path (use n_feat % 32 == 0)")
        wp = packed_fwd(w, dtype, Cp)
        M = B * Ho * Wo
        geom = dict(dtype=dtype, B=B, Hi=Hi, Wi=Wi, C1=C1, C2=C2, Hq=Ho, Wq=Wo, sy=s, sx=s, T=T, KW=spec.kw, ty=1, tx=1,
                    oy0=-p, ox0=-p, Ho=Ho, Wo=Wo, N=N)
        if bcast:
            geom["in2_batch"] = x2.shape[0]
        bn = spec.bn
        ctx.spec, ctx.geom, ctx.has_bn, ctx.C2, ctx.Cin = spec, geom, bn is not None, C2, Cin
        ctx.bias_present = b is not None
        if bn is None:
            if spec.out_nchw:
                out = _empty((B, N, Ho, Wo), torch.float32, x)
            else:
                out = _empty((B, Ho, Wo, N), dtype, x)
            if spec.act != ACT_NONE and need_grad:
                raise L.DmError("conv without BN: fused activation has no backward; use a separate activation")
            _conv_call(x, x2, ptr(wp), T * Cp, out, shift=b, act=spec.act, out_nchw=spec.out_nchw, **geom)
            ctx.save_for_backward(x, x2, w)
            return out
        train = bn.training
        if not train and not need_grad:
            sc, sh = _bn_eval_fold(spec, b)
            out = _empty((B, Ho, Wo, N), dtype, x)
            _conv_call(x, x2, ptr(wp), T * Cp, out, scale=sc, shift=sh, act=spec.act, **geom)
            return out
        z = _empty((B, Ho, Wo, N), dtype, x)
        mean, rstd = _empty((N,), torch.float32, x), _empty((N,), torch.float32, x)
        out = _empty((B, Ho, Wo, N), dtype, x)
        slots = BN_SLOTS if (train and N % (4 if dtype == torch.float32 else 8) == 0 and N <= BN_SLOTS_MAX_C) else 0
        if train and slots:
            # the conv epilogue adds its per-tile column sums into [slots][N] accumulators; the normalise + activation kernel folds
            # them in its prologue (and updates the running statistics): no finalize launch in between
            st = _gzeros((2, slots, 2 * N), x)               # [2][slots][N] doubles
            _conv_call(x, x2, ptr(wp), T * Cp, z, shift=b, psum=st[0], psq=st[1], stat_slots=slots, **geom)
            call("dm_bn_act_fwd_slots", ptr(z), ptr(out), dt(dtype), M, N, ptr(st[0]), ptr(st[1]), slots, BN_EPS, bn.momentum, ptr(gamma), ptr(beta),
                 spec.act, ptr(mean), ptr(rstd), ptr(bn.running_mean), ptr(bn.running_var))
            if PROFILE_META is not None:           # algorithmic bytes of the streaming pass: read z, write a
                PROFILE_META.append(("bn_fwd", 2.0 * M * N * z.element_size(), f"M{M} N{N}"))
            spec.nbt_pending += 1
            bn._stat_epoch = getattr(bn, "_stat_epoch", 0) + 1
        else:
            if train:
                nblk = (M + 127) // 128
                psum, psq = _empty((nblk, N), torch.float32, x), _empty((nblk, N), torch.float32, x)
                _conv_call(x, x2, ptr(wp), T * Cp, z, shift=b, psum=psum, psq=psq, **geom)
                call("dm_bn_finalize", ptr(psum), ptr(psq), nblk, M, N, BN_EPS, bn.momentum, ptr(mean), ptr(rstd),
                     ptr(bn.running_mean), ptr(bn.running_var))
                spec.nbt_pending += 1
                bn._stat_epoch = getattr(bn, "_stat_epoch", 0) + 1
            else:  # eval mode with autograd: normalise with the running statistics, unfused
                _conv_call(x, x2, ptr(wp), T * Cp, z, shift=b, **geom)
                mean, rstd = _running_stats(bn, N)
            call("dm_bn_act_fwd", ptr(z), ptr(out), dt(dtype), M, N, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), spec.act)
            if PROFILE_META is not None:
                PROFILE_META.append(("bn_fwd", 2.0 * M * N * z.element_size(), f"M{M} N{N}"))
        ctx.train = train
        ctx.save_for_backward(x, x2, w, z, mean, rstd, gamma, beta)
        return out

    @staticmethod
    def backward(ctx, g):
        spec, geom = ctx.spec, ctx.geom
        dtype = geom["dtype"]
        B, Hi, Wi, C1, C2 = geom["B"], geom["Hi"], geom["Wi"], geom["C1"], ctx.C2
        Ho, Wo, N, T = geom["Ho"], geom["Wo"], geom["N"], geom["T"]
        M = B * Ho * Wo
        dgamma = dbeta = None
        if ctx.has_bn:
            x, x2, w, z, mean, rstd, gamma, beta = ctx.saved_tensors
            g = g.contiguous()
            dbeta, dgamma = _empty((N,), torch.float32, g), _empty((N,), torch.float32, g)
            dz = _empty(z.shape, dtype, g)
            if PROFILE_META is not None:           # reduce: read z, dy; apply: read z, dy, write dz
                PROFILE_META.append(("bn_bwd_reduce", 2.0 * M * N * z.element_size(), f"M{M} N{N}"))
                PROFILE_META.append(("bn_bwd_apply", 3.0 * M * N * z.element_size(), f"M{M} N{N}"))
            slots = BN_SLOTS if (ctx.train and N % (4 if dtype == torch.float32 else 8) == 0 and N <= BN_SLOTS_MAX_C) else 0
            if slots:      # sums of g and g*xhat as [slots][N] accumulators folded by the apply kernel (which also emits dbeta / dgamma)
                pp = _gzeros((2, slots, 2 * N), g)           # [2][slots][N] doubles
                call("dm_bn_act_bwd_reduce_slots", ptr(z), ptr(g), dt(dtype), M, N, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), spec.act,
                     ptr(pp[0]), ptr(pp[1]), slots)
                call("dm_bn_act_bwd_apply_slots", ptr(z), ptr(g), ptr(dz), dt(dtype), M, N, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
                     spec.act, ptr(pp[0]), ptr(pp[1]), slots, ptr(dbeta), ptr(dgamma))
            else:
                nblk = L.colstat_blocks(M)
                p1, p2 = _empty((nblk, N), torch.float32, g), _empty((nblk, N), torch.float32, g)
                call("dm_bn_act_bwd_reduce", ptr(z), ptr(g), dt(dtype), M, N, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), spec.act,
                     ptr(p1), ptr(p2))
                call("dm_col_reduce2", ptr(p1), ptr(p2), nblk, N, ptr(dbeta), ptr(dgamma))
                if ctx.train:
                    s1, s2 = dbeta, dgamma
                else:
                    s1 = s2 = _gzeros((N,), g)
                call("dm_bn_act_bwd_apply", ptr(z), ptr(g), ptr(dz), dt(dtype), M, N, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
                     spec.act, ptr(s1), ptr(s2))
            ldy = N
        else:
            x, x2, w = ctx.saved_tensors
            if spec.out_nchw:
                ldy = (N + 7) // 8 * 8
                dz = _empty((B, Ho, Wo, ldy), dtype, g)
                call("dm_nchw_to_nhwc", ptr(g.contiguous().float()), ptr(dz), dt(dtype), B, N, Ho, Wo, ldy, 1)
            else:
                dz, ldy = g.contiguous(), N
        needs = ctx.needs_input_grad
        Cin, Cp = ctx.Cin, C1 + C2
        s, p, kh, kw = spec.stride, spec.pad, spec.kh, spec.kw
        dx = dx2 = dw = db = None
        # ---- input gradient: gather-GEMM of dz with transposed weights, one launch per output parity class
        if needs[0] or (x2 is not None and needs[1]):
            targets = [(0, C1, needs[0])] + ([(C1, C2, needs[1])] if x2 is not None else [])
            outs = []
            stash = ctx.fork.take() if ctx.fork is not None else None
            if stash is not None and (not needs[0] or tuple(stash.shape) != (B, Hi, Wi, C1) or stash.dtype != dtype):
                raise L.DmError("conv backward: the stashed gradient of the forked input does not match it")
            for (c_lo, c_n, need) in targets:
                if not need:
                    outs.append(None)
                    continue
                dxi = _empty((B, Hi, Wi, c_n), dtype, g)
                add_i = stash if c_lo == 0 else None
                if (kh % s) or (kw % s):
                    raise L.DmError("conv backward: kernel size must be a multiple of the stride")
                if c_lo + c_n > Cin:       # padded stem input: its gradient is never needed
                    raise L.DmError("conv backward: gradient w.r.t. a channel-padded input is not supported")
                if Hi % s or Wi % s:
                    raise L.DmError("conv backward: strided conv needs input size divisible by the stride")
                launches = []
                for py in range(s):
                    for px in range(s):
                        ky0, kx0 = (py + p) % s, (px + p) % s
                        taps = [(ky0 + s * a) * kw + (kx0 + s * bq) for a in range(kh // s) for bq in range(kw // s)]
                        wt = packed_T(w, dtype, taps, ldy)
                        # rows c_lo .. c_lo+c_n of the [C][Tt][ldy] pack
                        row_bytes = len(taps) * ldy * wt.element_size()
                        launches.append((dz, None, wt.data_ptr() + c_lo * row_bytes, len(taps) * ldy, dxi, dict(
                            dtype=dtype, B=B, Hi=Ho, Wi=Wo, C1=ldy, C2=0, Hq=Hi // s, Wq=Wi // s, sy=1, sx=1, T=len(taps), KW=kw // s, ty=-1, tx=-1,
                            oy0=(py + p - ky0) // s, ox0=(px + p - kx0) // s, Ho=Hi, Wo=Wi, osy=s, osx=s, ooy=py, oox=px, N=c_n, addend=add_i)))
                if len(launches) == 4:          # stride 2: the four output-parity classes in one call (one launch on the four-tap halo kernel)
                    _conv_parity4_call(launches)
                else:
                    for (a1, a2, wp_, ldw_, o_, kw_) in launches:
                        _conv_call(a1, a2, wp_, ldw_, o_, **kw_)
                outs.append(dxi)
            dx = outs[0]
            dx2 = outs[1] if x2 is not None else None
        # ---- weight / bias gradient
        if needs[2] or (ctx.bias_present and needs[3]):
            main = getattr(w, "main_grad", None)
            wg_geom = dict(dtype=dtype, B=B, Hi=Hi, Wi=Wi, C1=C1, C2=C2, Hq=Ho, Wq=Wo, sy=s, sx=s, T=T, KW=kw, ty=1, tx=1,
                           oy0=-p, ox0=-p, Ho=Ho, Wo=Wo, N=N, ldy=ldy, ldw=T * Cp)
            if ctx.bias_present:
                db = _gzeros((N,), g)
            if Cp == Cin:
                if main is not None:
                    tgt = main
                else:
                    tgt = _zeros((N, kh, kw, Cin), torch.float32, g)
                # FusedAdamW.zero_grad leaves the gradient ranges of the layers on the halo weight-gradient kernel un-zeroed and marks
                # them fresh: the first launch into such a range overwrites (its reduce launch writes instead of read-modify-writing)
                fresh = main is not None and getattr(w, "_dm_fresh", False)
                _wgrad_call(dz, x, x2, tgt, db, overwrite=fresh, **wg_geom)
                if main is not None:
                    w._dm_fresh = False
                    w._dm_halo_ok = getattr(w, "_dm_halo_ok", True) and L.load().dm_last_wgrad_path() == 1
                dw = None if main is not None else tgt.permute(0, 3, 1, 2)
                if main is not None and ON_WGRAD is not None:
                    ON_WGRAD(w)
            else:
                tmp = _zeros((N, T, Cp), torch.float32, g)
                _wgrad_call(dz, x, x2, tmp, db, **wg_geom)
                if main is not None:
                    call("dm_unpad_dw", ptr(tmp), ptr(main), N, T, Cin, Cp, 1)
                    if ON_WGRAD is not None:
                        ON_WGRAD(w)
                else:
                    tgt = _empty((N, kh, kw, Cin), torch.float32, g)
                    call("dm_unpad_dw", ptr(tmp), ptr(tgt), N, T, Cin, Cp, 0)
                    dw = tgt.permute(0, 3, 1, 2)
        return dx, dx2, dw, db, dgamma, dbeta, None, None, None


def conv_bn_act(x, x2, conv, bn, spec, fork=None):
    """conv: holder with .weight/.bias (nn.Conv2d); bn: nn.BatchNorm2d holder or None."""
    gamma = bn.weight if bn is not None else None
    beta = bn.bias if bn is not None else None
    need_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (x, x2, conv.weight, conv.bias, gamma, beta))
    return ConvBnAct.apply(x, x2, conv.weight, conv.bias, gamma, beta, spec, need_grad, fork)


# ------------------------------------------------------------------------------------------------
# ConvTranspose2d with kernel == stride (non-overlapping): a plain GEMM + scatter
# ------------------------------------------------------------------------------------------------
class ConvTransposeKS(torch.autograd.Function):
    """out[b, y*k+ky, x*k+kx, co] = sum_ci in[b,y,x,ci] W[ci,co,ky,kx] + bias[co]   (new_scripy.py:298, MNIST_script.py:88,141)"""

    @staticmethod
    def forward(ctx, x, w, b, dtype):
        L.require_device(x, w)
        B, h, wd, Cin = x.shape
        Cout, k = w.shape[1], w.shape[2]
        if x.dtype != dtype:
            xin = _empty(x.shape, dtype, x)
            call("dm_cast", ptr(x.contiguous()), ptr(xin), dt(x.dtype), dt(dtype), x.numel())
        else:
            xin = x.contiguous()
        _cl(w)                                        # physical [Cin][k][k][Cout]
        T = k * k

        def build():                                  # forward pack: [(t,co)][ci]
            o = _empty((T * Cout, Cin), dtype, w)
            call("dm_pack_wT", ptr(w), ptr(o), dt(dtype), Cin, 1, T * Cout, 1, None, Cin)
            return o
        wf = _cached(("ctf", dtype), w, build)
        out = _empty((B, h * k, wd * k, Cout), dtype, x)
        bias_rep = b
        if h == 1 and wd == 1:
            def build_b():
                return b.detach().repeat(T).contiguous()
            bias_rep = _cached(("ctb", T), b, build_b)
            _conv_call(xin, None, ptr(wf), Cin, out, dtype=dtype, B=B, Hi=1, Wi=1, C1=Cin, C2=0, Hq=1, Wq=1, sy=1, sx=1, T=1, KW=1,
                       ty=1, tx=1, oy0=0, ox0=0, Ho=1, Wo=1, N=T * Cout, shift=bias_rep)
        else:
            for t in range(T):
                _conv_call(xin, None, wf.data_ptr() + t * Cout * Cin * wf.element_size(), Cin, out, dtype=dtype, B=B, Hi=h, Wi=wd,
                           C1=Cin, C2=0, Hq=h, Wq=wd, sy=1, sx=1, T=1, KW=1, ty=1, tx=1, oy0=0, ox0=0, Ho=h * k, Wo=wd * k,
                           osy=k, osx=k, ooy=t // k, oox=t % k, N=Cout, shift=b)
        ctx.save_for_backward(xin, w)
        ctx.meta = (B, h, wd, Cin, Cout, k, dtype, x.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        xin, w = ctx.saved_tensors
        B, h, wd, Cin, Cout, k, dtype, xdtype = ctx.meta
        g = g.contiguous()
        T = k * k
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wd_ = packed_fwd(w, dtype, Cout)          # [Cin][T][Cout]: rows = ci, K = (t, co)
            dxi = _empty((B, h, wd, Cin), dtype, g)
            _conv_call(g, None, ptr(wd_), T * Cout, dxi, dtype=dtype, B=B, Hi=h * k, Wi=wd * k, C1=Cout, C2=0, Hq=h, Wq=wd, sy=k, sx=k,
                       T=T, KW=k, ty=1, tx=1, oy0=0, ox0=0, Ho=h, Wo=wd, N=Cin)
            if xdtype != dtype:
                dx = _empty(dxi.shape, xdtype, g)
                call("dm_cast", ptr(dxi), ptr(dx), dt(dtype), dt(xdtype), dxi.numel())
            else:
                dx = dxi
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            main = getattr(w, "main_grad", None)
            tgt = main if main is not None else _zeros((Cin, k, k, Cout), torch.float32, g)
            # "dy" role = the ConvT input (N = Cin), "in" role = g gathered with taps (ky,kx), stride k
            _wgrad_call(xin, g, None, tgt, None, dtype=dtype, B=B, Hi=h * k, Wi=wd * k, C1=Cout, C2=0, Hq=h, Wq=wd, sy=k, sx=k, T=T, KW=k,
                        ty=1, tx=1, oy0=0, ox0=0, Ho=h, Wo=wd, N=Cin, ldy=Cin, ldw=T * Cout)
            dw = None if main is not None else tgt.permute(0, 3, 1, 2)
            if main is not None and ON_WGRAD is not None:
                ON_WGRAD(w)
            # bias gradient: column sums of g over all pixels
            M = B * h * k * wd * k
            nblk = L.colstat_blocks(M)
            p1, p2 = _empty((nblk, Cout), torch.float32, g), _empty((nblk, Cout), torch.float32, g)
            call("dm_col_stats", ptr(g), dt(dtype), M, Cout, ptr(p1), ptr(p2))
            db = _empty((Cout,), torch.float32, g)
            call("dm_col_reduce", ptr(p1), nblk, Cout, ptr(db), 0)
        return dx, dw, db, None


# ------------------------------------------------------------------------------------------------
# GroupNorm + activation
# ------------------------------------------------------------------------------------------------
class GroupNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, act):
        L.require_device(x)
        x = x.contiguous()
        B, H, W, Cc = x.shape
        y = _empty(x.shape, x.dtype, x)
        mean, rstd = _empty((B * groups,), torch.float32, x), _empty((B * groups,), torch.float32, x)
        L.ensure_workspace()                     # large images: per-slab partial sums
        call("dm_gn_act_fwd", ptr(x), ptr(y), dt(x), B, H * W, Cc, groups, BN_EPS, ptr(gamma), ptr(beta), act, ptr(mean), ptr(rstd))
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.meta = (groups, act)
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        groups, act = ctx.meta
        B, H, W, Cc = x.shape
        g = g.contiguous()
        dx = _empty(x.shape, x.dtype, x)
        dgamma, dbeta = _gzeros((Cc,), x), _gzeros((Cc,), x)
        L.ensure_workspace()
        call("dm_gn_act_bwd", ptr(x), ptr(g), ptr(dx), dt(x), B, H * W, Cc, groups, ptr(gamma), ptr(beta), act, ptr(mean), ptr(rstd),
             ptr(dgamma), ptr(dbeta))
        return dx, dgamma, dbeta, None, None


# ------------------------------------------------------------------------------------------------
# fp32 dense layers / activations on small matrices
# ------------------------------------------------------------------------------------------------
def _lin_fwd(x, w, b, y, act=ACT_NONE):
    """y[M][N] = act(x[M][K] w[N][K]^T + b), fp32 (dense.hip: one wave-level MFMA pass per launch; odd widths: the plain sgemm)."""
    M, K = x.shape
    call("dm_linear_fwd", ptr(x), ptr(w), ptr(b), ptr(y), M, K, w.shape[0], act)


def _lin_bwd(x, w, g, dx, dw, db):
    """dx = g w (overwritten), dw += g^T x, db += colsum(g); any of dx/dw/db may be None."""
    M, K = x.shape
    N = w.shape[0]
    if dw is None and db is not None:
        dw = _gzeros((N, K), x)
    call("dm_linear_bwd", ptr(x), ptr(w), ptr(g), ptr(dx), ptr(dw), ptr(db), M, K, N)


class Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, fork=None):
        ctx.fork = fork
        L.require_device(x, w)
        x, w = x.contiguous(), w.contiguous()
        y = _empty((x.shape[0], w.shape[0]), torch.float32, x)
        _lin_fwd(x, w, b, y)
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.contiguous()
        M, K = x.shape
        N = w.shape[0]
        dx = _empty((M, K), torch.float32, x) if ctx.needs_input_grad[0] else None
        dw = _gzeros((N, K), x) if ctx.needs_input_grad[1] else None
        db = _gzeros((N,), x) if (ctx.has_b and ctx.needs_input_grad[2]) else None
        _lin_bwd(x, w, g, dx, dw, db)
        stash = ctx.fork.take() if ctx.fork is not None else None
        if stash is not None and dx is None:
            raise L.DmError("Linear backward: a gradient was stashed for the forked input, but the input needs no gradient here")
        if stash is not None:
            call("dm_add", ptr(dx), ptr(stash.reshape(dx.shape).contiguous()), ptr(dx), L.DM_F32, dx.numel())
        return dx, dw, db, None


def linear_act(x, w, b, act):
    """Inference-only dense layer with the activation in its epilogue (no autograd node)."""
    L.require_device(x, w)
    x, w = x.contiguous(), w.contiguous()
    y = _empty((x.shape[0], w.shape[0]), torch.float32, x)
    _lin_fwd(x, w, b, y, act)
    return y


def linear(x, w, b=None, fork=None):
    """x (..., K) fp32 -> (..., N); w may be a 1x1-conv weight (N, K, 1, 1)."""
    w2 = w.reshape(w.shape[0], -1) if w.dim() == 4 else w
    lead = x.shape[:-1]
    y = Linear.apply(x.reshape(-1, x.shape[-1]), w2, b, fork)
    return y.reshape(*lead, w2.shape[0])


class Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        L.require_device(x)
        x = x.contiguous()
        y = _empty(x.shape, torch.float32, x)
        call("dm_act_fwd", ptr(x), ptr(y), x.numel(), act)
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = _empty(x.shape, torch.float32, x)
        call("dm_act_bwd", ptr(x), ptr(g.contiguous()), ptr(dx), x.numel(), ctx.act)
        return dx, None


class BnActMatrix(torch.autograd.Function):
    """BatchNorm (+act) over the rows of an fp32 [M][C] matrix — CoordAttn's bn1_h / bn1_w on strips."""

    @staticmethod
    def forward(ctx, z, gamma, beta, bn, spec, act):
        L.require_device(z)
        shape = z.shape
        z = z.contiguous().reshape(-1, shape[-1])
        M, Cc = z.shape
        mean, rstd = _empty((Cc,), torch.float32, z), _empty((Cc,), torch.float32, z)
        train = bn.training
        if train:
            nblk = L.colstat_blocks(M)
            p1, p2 = _empty((nblk, Cc), torch.float32, z), _empty((nblk, Cc), torch.float32, z)
            call("dm_col_stats", ptr(z), L.DM_F32, M, Cc, ptr(p1), ptr(p2))
            call("dm_bn_finalize", ptr(p1), ptr(p2), nblk, M, Cc, BN_EPS, bn.momentum, ptr(mean), ptr(rstd), ptr(bn.running_mean),
                 ptr(bn.running_var))
            spec.nbt_pending += 1
        else:
            mean, rstd = _running_stats(bn, Cc)
        y = _empty(z.shape, torch.float32, z)
        call("dm_bn_act_fwd", ptr(z), ptr(y), L.DM_F32, M, Cc, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), act)
        ctx.save_for_backward(z, mean, rstd, gamma, beta)
        ctx.meta = (train, act, shape)
        return y.reshape(shape)

    @staticmethod
    def backward(ctx, g):
        z, mean, rstd, gamma, beta = ctx.saved_tensors
        train, act, shape = ctx.meta
        M, Cc = z.shape
        g = g.contiguous().reshape(M, Cc)
        nblk = L.colstat_blocks(M)
        p1, p2 = _empty((nblk, Cc), torch.float32, z), _empty((nblk, Cc), torch.float32, z)
        call("dm_bn_act_bwd_reduce", ptr(z), ptr(g), L.DM_F32, M, Cc, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), act, ptr(p1), ptr(p2))
        dbeta, dgamma = _empty((Cc,), torch.float32, z), _empty((Cc,), torch.float32, z)
        call("dm_col_reduce2", ptr(p1), ptr(p2), nblk, Cc, ptr(dbeta), ptr(dgamma))
        dz = _empty(z.shape, torch.float32, z)
        s1, s2 = (dbeta, dgamma) if train else (_gzeros((Cc,), z),) * 2
        call("dm_bn_act_bwd_apply", ptr(z), ptr(g), ptr(dz), L.DM_F32, M, Cc, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), act,
             ptr(s1), ptr(s2))
        return dz.reshape(shape), dgamma, dbeta, None, None, None


# ------------------------------------------------------------------------------------------------
# SE + residual combine (one node): out = (res + x2 * sigmoid(W2 gelu(W1 mean_hw(x2)))) / 1.414
# ------------------------------------------------------------------------------------------------
class SeResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2, res, w1, w2, inv, need_grad=True):
        L.require_device(x2, res)
        x2, res = x2.contiguous(), res.contiguous()
        B, H, W, Cc = x2.shape
        out = _empty(x2.shape, x2.dtype, x2)
        if w1 is None:                      # MNIST block: no SE, plain residual
            call("dm_scale_residual_fwd", ptr(x2), ptr(res), None, ptr(out), dt(x2), B, H * W, Cc, inv)
            ctx.save_for_backward()
            ctx.meta = (None, inv, x2.shape, x2.dtype)
            return out
        R = w1.shape[0]
        L.ensure_workspace()                     # the whole-image pools split their rows over workgroups through it
        w1, w2 = w1.contiguous(), w2.contiguous()
        if FUSED_CHAINS and Cc % 4 == 0 and R % 4 == 0 and R <= 128:
            # pooling partials -> (fold, fc, GELU, fc, sigmoid) in one kernel per 16 samples (chain.hip)
            sg = _empty((B, Cc), torch.float32, x2)
            if need_grad:
                y, hid, gh = _empty((B, Cc), torch.float32, x2), _empty((B, R), torch.float32, x2), _empty((B, R), torch.float32, x2)
            else:
                y = hid = gh = None
            call("dm_se_fwd", ptr(x2), dt(x2), B, H * W, Cc, ptr(w1), ptr(w2), R, ptr(y), ptr(hid), ptr(gh), ptr(sg))
            call("dm_scale_residual_fwd", ptr(x2), ptr(res), ptr(sg), ptr(out), dt(x2), B, H * W, Cc, inv)
            if need_grad:
                ctx.save_for_backward(x2, y, hid, gh, None, sg, w1, w2)
            ctx.meta = (R, inv, x2.shape, x2.dtype)
            ctx.fused = True
            return out
        ctx.fused = False
        y = _empty((B, Cc), torch.float32, x2)
        call("dm_pool_hw", ptr(x2), dt(x2), B, H * W, Cc, ptr(y))
        hid, gh = _empty((B, R), torch.float32, x2), _empty((B, R), torch.float32, x2)
        logit, sg = _empty((B, Cc), torch.float32, x2), _empty((B, Cc), torch.float32, x2)
        if not need_grad:                        # inference (the sampler): activations in the dense epilogues, nothing kept
            _lin_fwd(y, w1, None, gh, ACT_GELU)
            _lin_fwd(gh, w2, None, sg, ACT_SIGMOID)
        else:
            _lin_fwd(y, w1, None, hid)
            call("dm_act_fwd", ptr(hid), ptr(gh), B * R, ACT_GELU)
            _lin_fwd(gh, w2, None, logit)
            call("dm_act_fwd", ptr(logit), ptr(sg), B * Cc, ACT_SIGMOID)
        call("dm_scale_residual_fwd", ptr(x2), ptr(res), ptr(sg), ptr(out), dt(x2), B, H * W, Cc, inv)
        ctx.save_for_backward(x2, y, hid, gh, logit, sg, w1, w2)
        ctx.meta = (R, inv, x2.shape, x2.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        R, inv, shape, dtype = ctx.meta
        B, H, W, Cc = shape
        g = g.contiguous()
        dx2, dres = _empty(shape, dtype, g), _empty(shape, dtype, g)
        if R is None:
            call("dm_scale_residual_bwd_apply", ptr(g), None, None, ptr(dx2), ptr(dres), dt(dtype), B, H * W, Cc, inv)
            return dx2, dres, None, None, None, None
        x2, y, hid, gh, logit, sg, w1, w2 = ctx.saved_tensors
        f = lambda *s: _empty(s, torch.float32, g)
        dw1, dw2 = _gzeros((R, Cc), g), _gzeros((Cc, R), g)
        L.ensure_workspace()
        if getattr(ctx, "fused", False):
            dlogit, dhid, dy = f(B, Cc), f(B, R), f(B, Cc)
            call("dm_se_bwd", ptr(g), ptr(x2), dt(dtype), B, H * W, Cc, inv, ptr(sg), ptr(hid), ptr(gh), ptr(y), ptr(w1), ptr(w2), R,
                 ptr(dlogit), ptr(dhid), ptr(dy), ptr(dw1), ptr(dw2))
            call("dm_scale_residual_bwd_apply", ptr(g), ptr(sg), ptr(dy), ptr(dx2), ptr(dres), dt(dtype), B, H * W, Cc, inv)
            return dx2, dres, dw1, dw2, None, None
        dsg, dlogit, dgh, dhid, dy = f(B, Cc), f(B, Cc), f(B, R), f(B, R), f(B, Cc)
        call("dm_scale_residual_bwd_reduce", ptr(g), ptr(x2), dt(dtype), B, H * W, Cc, inv, ptr(dsg))
        call("dm_act_bwd", ptr(logit), ptr(dsg), ptr(dlogit), B * Cc, ACT_SIGMOID)
        _lin_bwd(gh, w2, dlogit, dgh, dw2, None)
        call("dm_act_bwd", ptr(hid), ptr(dgh), ptr(dhid), B * R, ACT_GELU)
        _lin_bwd(y, w1, dhid, dy, dw1, None)
        call("dm_scale_residual_bwd_apply", ptr(g), ptr(sg), ptr(dy), ptr(dx2), ptr(dres), dt(dtype), B, H * W, Cc, inv)
        return dx2, dres, dw1, dw2, None, None


# ------------------------------------------------------------------------------------------------
# Coordinate attention pieces
# ------------------------------------------------------------------------------------------------
class PoolStrips(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fork=None):
        ctx.fork = fork
        L.require_device(x)
        x = x.contiguous()
        B, H, W, Cc = x.shape
        xh, xw = _empty((B, H, Cc), torch.float32, x), _empty((B, W, Cc), torch.float32, x)
        call("dm_ca_pool_fwd", ptr(x), dt(x), B, H, W, Cc, ptr(xh), ptr(xw))
        ctx.meta = (x.shape, x.dtype)
        return xh, xw

    @staticmethod
    def backward(ctx, gh, gw):
        shape, dtype = ctx.meta
        B, H, W, Cc = shape
        dx = _empty(shape, dtype, gh)
        stash = ctx.fork.take() if ctx.fork is not None else None      # CaGate's gradient of x (new_scripy.py:138-140), added in the same pass
        call("dm_ca_pool_bwd", ptr(gh.contiguous()), ptr(gw.contiguous()), ptr(stash), ptr(dx), dt(dtype), B, H, W, Cc)
        return dx, None


class SigMix(torch.autograd.Function):
    """xo = x + sigmoid(gamma) * y  (new_scripy.py:122-126)"""

    @staticmethod
    def forward(ctx, x, y, gamma):
        x, y = x.contiguous(), y.contiguous()
        xo = _empty(x.shape, torch.float32, x)
        call("dm_sigmix_fwd", ptr(x), ptr(y), ptr(gamma), ptr(xo), x.numel())
        ctx.save_for_backward(y, gamma)
        return xo

    @staticmethod
    def backward(ctx, g):
        y, gamma = ctx.saved_tensors
        g = g.contiguous()
        dy = _empty(y.shape, torch.float32, y)
        dgamma = _gzeros((1,), y)
        call("dm_sigmix_bwd", ptr(g), ptr(y), ptr(gamma), ptr(dy), ptr(dgamma), y.numel())
        return g, dy, dgamma


class CaGate(torch.autograd.Function):
    """out = x * (al*sigmoid(lh) + be*sigmoid(lw)) with al/be the normalised sigmoid(alpha/beta) (new_scripy.py:128-140)"""

    @staticmethod
    def forward(ctx, x, lh, lw, alpha, beta):
        x, lh, lw = x.contiguous(), lh.contiguous(), lw.contiguous()
        B, H, W, Cc = x.shape
        out = _empty(x.shape, x.dtype, x)
        call("dm_ca_gate_fwd", ptr(x), ptr(lh), ptr(lw), ptr(alpha), ptr(beta), ptr(out), dt(x), B, H, W, Cc)
        ctx.save_for_backward(x, lh, lw, alpha, beta)
        return out

    @staticmethod
    def backward(ctx, g):
        x, lh, lw, alpha, beta = ctx.saved_tensors
        B, H, W, Cc = x.shape
        g = g.contiguous()
        dx = _empty(x.shape, x.dtype, x)
        dlh, dlw = _empty(lh.shape, torch.float32, x), _empty(lw.shape, torch.float32, x)
        dab = _gzeros((4,), x)
        call("dm_ca_gate_bwd", ptr(x), ptr(g), ptr(lh), ptr(lw), ptr(alpha), ptr(beta), ptr(dx), ptr(dlh), ptr(dlw), ptr(dab), dt(x),
             B, H, W, Cc)
        return dx, dlh, dlw, dab[0:1], dab[1:2]


def ca_chain_lds(H, W, R):
    """Dynamic LDS bytes of ca_mix / ca_bwd_mix (chain.hip:ca_geometry): three strip matrices of (Hp + Wp) x (R + 4) floats + 8 R."""
    hp, wp = -(-H // 16) * 16, -(-W // 16) * 16
    return (3 * (hp + wp) * (R + 4) + 8 * R) * 4


def ca_chain_ok(C_, R, H=None, W=None):
    """Shapes the fused CoordAttn strip chain (chain.hip) takes: C % 4 == 0, R % 4 == 0, R <= 128, strips inside 120 KiB of LDS — and
    inside 42 KiB (+ 22 KiB static = 64 KiB per workgroup) when the library runs its small-LDS kernels because the GPU is shared
    between processes (_lib.device_guard / parallel.guard_shared_device -> conv variant 2)."""
    if not (FUSED_CHAINS and C_ % 4 == 0 and R % 4 == 0 and 0 < R <= 128):
        return False
    if H is None:
        return True
    need = ca_chain_lds(H, W, R)
    if need > 120 * 1024:
        return False
    return not (L.load().dm_get_conv_variant() == 2 and need > (64 - 22) * 1024)


class CaChain(torch.autograd.Function):
    """(x_h, x_w) -> (l_h, l_w): conv1 + BatchNorm + GELU on both strips, the two cross projections, the sigmoid-gated mix with
    F.adaptive_avg_pool2d between strips of different length, conv_h / conv_w (new_scripy.py:105-129) — two launches forward, two
    backward (dm_ca_chain_fwd / dm_ca_chain_bwd) instead of one per torch.nn op."""
    PARAMS = ("conv1_h.weight", "conv1_h.bias", "conv1_w.weight", "conv1_w.bias", "bn1_h.weight", "bn1_h.bias", "bn1_w.weight", "bn1_w.bias",
              "h2w_proj.weight", "h2w_proj.bias", "w2h_proj.weight", "w2h_proj.bias", "gamma_h", "gamma_w", "conv_h.weight", "conv_h.bias",
              "conv_w.weight", "conv_w.bias")

    @staticmethod
    def forward(ctx, xh, xw, mod, need_grad, *prm):
        L.require_device(xh, xw)
        xh, xw = xh.contiguous(), xw.contiguous()
        B, H, Cc = xh.shape
        W = xw.shape[1]
        (w1h, b1h, w1w, b1w, gh_, bh_, gw_, bw_, whw, bhw, wwh, bwh, gam_h, gam_w, wch, bch, wcw, bcw) = prm
        R = w1h.shape[0]
        mats = [t.reshape(t.shape[0], -1).contiguous() for t in (w1h, w1w, whw, wwh, wch, wcw)]
        d = L.DmCaChain()
        d.B, d.H, d.W, d.C, d.R = B, H, W, Cc, R
        train = mod.bn1_h.training
        if train != mod.bn1_w.training:
            raise L.DmError("CoordAttn: bn1_h and bn1_w must be in the same mode")
        d.train, d.save = int(train), int(bool(need_grad))
        d.eps, d.momentum = BN_EPS, float(mod.bn1_h.momentum)
        f = lambda *s_: _empty(s_, torch.float32, xh)
        zh, zw, lh, lw = f(B * H, R), f(B * W, R), f(B, H, Cc), f(B, W, Cc)
        saved = [f(R), f(R), f(R), f(R), f(B * H, R), f(B * W, R), f(B * H, R), f(B * W, R)] if need_grad else [None] * 8
        stat = f((-(-B * H // 16) + -(-B * W // 16)) * 2 * R) if train else None
        d.xh, d.xw = ptr(xh), ptr(xw)
        d.w1h, d.b1h, d.w1w, d.b1w = ptr(mats[0]), ptr(b1h), ptr(mats[1]), ptr(b1w)
        d.bn_h_g, d.bn_h_b, d.bn_w_g, d.bn_w_b = ptr(gh_), ptr(bh_), ptr(gw_), ptr(bw_)
        d.rm_h, d.rv_h, d.rm_w, d.rv_w = ptr(mod.bn1_h.running_mean), ptr(mod.bn1_h.running_var), ptr(mod.bn1_w.running_mean), ptr(mod.bn1_w.running_var)
        d.whw, d.bhw, d.wwh, d.bwh = ptr(mats[2]), ptr(bhw), ptr(mats[3]), ptr(bwh)
        d.gam_h, d.gam_w = ptr(gam_h), ptr(gam_w)
        d.wch, d.bch, d.wcw, d.bcw = ptr(mats[4]), ptr(bch), ptr(mats[5]), ptr(bcw)
        d.zh, d.zw, d.lh, d.lw, d.stat = ptr(zh), ptr(zw), ptr(lh), ptr(lw), ptr(stat)
        (d.mean_h, d.rstd_h, d.mean_w, d.rstd_w, d.ah, d.aw, d.xhp, d.xwp) = [ptr(t) for t in saved]
        call("dm_ca_chain_fwd", C.byref(d))
        if train:
            for sp, bn in ((mod._sp_h, mod.bn1_h), (mod._sp_w, mod.bn1_w)):
                sp.nbt_pending += 1
                bn._stat_epoch = getattr(bn, "_stat_epoch", 0) + 1
        if need_grad:
            ctx.save_for_backward(xh, xw, zh, zw, *saved, *mats, gh_, bh_, gw_, bw_, gam_h, gam_w, bhw, bwh)
            ctx.meta = (B, H, W, Cc, R, train, float(mod.bn1_h.momentum), [t is not None for t in (b1h, b1w, bhw, bwh, bch, bcw)],
                        [tuple(t.shape) for t in (w1h, w1w, whw, wwh, wch, wcw)])
        return lh, lw

    @staticmethod
    def backward(ctx, dlh, dlw):
        (xh, xw, zh, zw, mean_h, rstd_h, mean_w, rstd_w, ah, aw, xhp, xwp, w1h, w1w, whw, wwh, wch, wcw, gh_, bh_, gw_, bw_, gam_h,
         gam_w, bhw, bwh) = ctx.saved_tensors
        B, H, W, Cc, R, train, mom, has_b, wshapes = ctx.meta
        dlh, dlw = dlh.contiguous(), dlw.contiguous()
        d = L.DmCaChain()
        d.B, d.H, d.W, d.C, d.R, d.train, d.save = B, H, W, Cc, R, int(train), 1
        d.eps, d.momentum = BN_EPS, mom
        f = lambda *s_: _empty(s_, torch.float32, xh)
        gz = lambda *s_: _gzeros(s_, xh)
        g_h, g_w, bnpart, dh2w, dw2h = f(B * H, R), f(B * W, R), f(2 * B * 2 * R), f(B * H, R), f(B * W, R)
        dxh, dxw = f(B, H, Cc), f(B, W, Cc)
        d_w1h, d_w1w, d_whw, d_wwh, d_wch, d_wcw = gz(R, Cc), gz(R, Cc), gz(R, R), gz(R, R), gz(Cc, R), gz(Cc, R)
        d_b = [gz(n) if h else None for n, h in zip((R, R, R, R, Cc, Cc), has_b)]
        d_bn = [f(R) for _ in range(4)]
        d_gam = gz(2)
        d.xh, d.xw, d.zh, d.zw = ptr(xh), ptr(xw), ptr(zh), ptr(zw)
        d.w1h, d.w1w, d.whw, d.wwh, d.wch, d.wcw = ptr(w1h), ptr(w1w), ptr(whw), ptr(wwh), ptr(wch), ptr(wcw)
        d.bhw, d.bwh = ptr(bhw), ptr(bwh)          # the projections are recomputed in the backward pass (d gamma needs them)
        d.bn_h_g, d.bn_h_b, d.bn_w_g, d.bn_w_b, d.gam_h, d.gam_w = ptr(gh_), ptr(bh_), ptr(gw_), ptr(bw_), ptr(gam_h), ptr(gam_w)
        d.mean_h, d.rstd_h, d.mean_w, d.rstd_w, d.ah, d.aw, d.xhp, d.xwp = [ptr(t) for t in (mean_h, rstd_h, mean_w, rstd_w, ah, aw, xhp, xwp)]
        d.dlh, d.dlw, d.gh, d.gw, d.bnpart, d.dh2w, d.dw2h, d.dxh, d.dxw = [ptr(t) for t in (dlh, dlw, g_h, g_w, bnpart, dh2w, dw2h, dxh, dxw)]
        d.d_w1h, d.d_w1w, d.d_whw, d.d_wwh, d.d_wch, d.d_wcw = [ptr(t) for t in (d_w1h, d_w1w, d_whw, d_wwh, d_wch, d_wcw)]
        d.d_b1h, d.d_b1w, d.d_bhw, d.d_bwh, d.d_bch, d.d_bcw = [ptr(t) for t in d_b]
        d.d_bn_h_g, d.d_bn_h_b, d.d_bn_w_g, d.d_bn_w_b = [ptr(t) for t in d_bn]
        d.d_gam = ptr(d_gam)
        call("dm_ca_chain_bwd", C.byref(d))
        dws = [t.reshape(sh) for t, sh in zip((d_w1h, d_w1w, d_whw, d_wwh, d_wch, d_wcw), wshapes)]
        return (dxh, dxw, None, None, dws[0], d_b[0], dws[1], d_b[1], d_bn[0], d_bn[1], d_bn[2], d_bn[3], dws[2], d_b[2], dws[3], d_b[3],
                d_gam[0:1], d_gam[1:2], dws[4], d_b[4], dws[5], d_b[5])


def ca_chain(xh, xw, mod):
    prm = []
    for name in CaChain.PARAMS:
        o = mod
        for part in name.split("."):
            o = getattr(o, part)
        prm.append(o)
    need = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in [xh, xw] + prm)
    return CaChain.apply(xh, xw, mod, need, *prm)


# ------------------------------------------------------------------------------------------------
# glue
# ------------------------------------------------------------------------------------------------
class UpCat(torch.autograd.Function):
    """bilinear x2 (align_corners=True) of cat(x1, x2) (new_scripy.py:242,251)"""

    @staticmethod
    def forward(ctx, x1, x2):
        L.require_device(x1, x2)
        x1, x2 = x1.contiguous(), x2.contiguous()
        B, H, W, C1 = x1.shape
        C2 = x2.shape[3]
        y = _empty((B, 2 * H, 2 * W, C1 + C2), x1.dtype, x1)
        if x2.shape[0] != B:                     # CFG sampler: the skip tensor was computed once for both halves of the batch
            if any(ctx.needs_input_grad):
                raise L.DmError("UpCat: a skip tensor with a smaller batch is inference-only")
            call("dm_upcat_fwd_bcast", ptr(x1), ptr(x2), ptr(y), dt(x1), B, x2.shape[0], H, W, C1, C2)
        else:
            call("dm_upcat_fwd", ptr(x1), ptr(x2), ptr(y), dt(x1), B, H, W, C1, C2)
        ctx.meta = (B, H, W, C1, C2, x1.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        B, H, W, C1, C2, dtype = ctx.meta
        g = g.contiguous()
        d1, d2 = _empty((B, H, W, C1), dtype, g), _empty((B, H, W, C2), dtype, g)
        call("dm_upcat_bwd", ptr(g), ptr(d1), ptr(d2), dt(dtype), B, H, W, C1, C2)
        return d1, d2


class Cat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2):
        x1, x2 = x1.contiguous(), x2.contiguous()
        B, H, W, C1 = x1.shape
        C2 = x2.shape[3]
        y = _empty((B, H, W, C1 + C2), x1.dtype, x1)
        call("dm_cat_fwd", ptr(x1), ptr(x2), ptr(y), dt(x1), B * H * W, C1, C2)
        ctx.meta = (B, H, W, C1, C2, x1.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        B, H, W, C1, C2, dtype = ctx.meta
        g = g.contiguous()
        d1, d2 = _empty((B, H, W, C1), dtype, g), _empty((B, H, W, C2), dtype, g)
        call("dm_cat_bwd", ptr(g), ptr(d1), ptr(d2), dt(dtype), B * H * W, C1, C2)
        return d1, d2


class Film(torch.autograd.Function):
    """y = cemb[b,c] * x + temb[b,c] (new_scripy.py:348-349)"""

    @staticmethod
    def forward(ctx, x, cemb, temb):
        x, cemb, temb = x.contiguous(), cemb.contiguous(), temb.contiguous()
        B, H, W, Cc = x.shape
        y = _empty(x.shape, x.dtype, x)
        call("dm_film_fwd", ptr(x), ptr(cemb), ptr(temb), ptr(y), dt(x), B, H * W, Cc)
        ctx.save_for_backward(x, cemb)
        return y

    @staticmethod
    def backward(ctx, g):
        x, cemb = ctx.saved_tensors
        B, H, W, Cc = x.shape
        g = g.contiguous()
        dx = _empty(x.shape, x.dtype, x)
        dce, dte = _empty((B, Cc), torch.float32, x), _empty((B, Cc), torch.float32, x)
        call("dm_film_bwd", ptr(x), ptr(g), ptr(cemb), ptr(dx), ptr(dce), ptr(dte), dt(x), B, H * W, Cc)
        return dx, dce, dte


class AvgPoolGelu(torch.autograd.Function):
    """AvgPool2d(k) + GELU -> fp32 (new_scripy.py:290)"""

    @staticmethod
    def forward(ctx, x, k, fork=None):
        ctx.fork = fork
        x = x.contiguous()
        B, H, W, Cc = x.shape
        y = _empty((B, H // k, W // k, Cc), torch.float32, x)
        call("dm_avgpool_gelu_fwd", ptr(x), ptr(y), dt(x), B, H, W, Cc, k)
        ctx.save_for_backward(x)
        ctx.k = k
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        B, H, W, Cc = x.shape
        if H % ctx.k or W % ctx.k:
            raise L.DmError("avgpool backward needs H, W divisible by k")
        dx = _empty(x.shape, x.dtype, x)
        call("dm_avgpool_gelu_bwd", ptr(x), ptr(g.contiguous()), ptr(dx), dt(x), B, H, W, Cc, ctx.k)
        stash = ctx.fork.take() if ctx.fork is not None else None
        if stash is not None:
            call("dm_add", ptr(dx), ptr(stash), ptr(dx), dt(x), dx.numel())
        return dx, None, None


class MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, H, W, Cc = x.shape
        y = _empty((B, H // 2, W // 2, Cc), x.dtype, x)
        call("dm_maxpool2_fwd", ptr(x), ptr(y), dt(x), B, H, W, Cc)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        B, H, W, Cc = x.shape
        dx = _empty(x.shape, x.dtype, x)
        call("dm_maxpool2_bwd", ptr(x), ptr(g.contiguous()), ptr(dx), dt(x), B, H, W, Cc)
        return dx


def onehot_mask(c, mask, n_classes, flip=False):
    L.require_device(c, mask)
    B = c.shape[0]
    out = _empty((B, n_classes), torch.float32, mask)
    call("dm_onehot_mask", ptr(c.contiguous()), ptr(mask.contiguous()), ptr(out), B, n_classes, int(flip))
    return out


def nchw_to_nhwc(x, dtype, cp, repeat=1):
    L.require_device(x)
    B, Cc, H, W = x.shape
    y = _empty((B * repeat, H, W, cp), dtype, x)
    call("dm_nchw_to_nhwc", ptr(x.contiguous()), ptr(y), dt(dtype), B, Cc, H, W, cp, repeat)
    return y


def nhwc_to_nchw(x, c):
    B, H, W, Cp = x.shape
    y = _empty((B, c, H, W), torch.float32, x)
    call("dm_nhwc_to_nchw", ptr(x.contiguous()), ptr(y), dt(x), B, c, H, W, Cp)
    return y


# ------------------------------------------------------------------------------------------------
# DDPM pieces
# ------------------------------------------------------------------------------------------------
def qsample(x, noise, ts, sqrtab, sqrtmab, dtype, cp):
    L.require_device(x, noise, ts)
    B, Cc, H, W = x.shape
    xt = _empty((B, H, W, cp), dtype, x)
    call("dm_qsample", ptr(x.contiguous()), ptr(noise.contiguous()), ptr(ts.contiguous()), ptr(sqrtab), ptr(sqrtmab), ptr(xt), dt(dtype),
         B, Cc, H, W, cp)
    return xt


class WeightedLoss(torch.autograd.Function):
    """mean(w*(n-p)^2) + feat_w*mean(|p*h - n*h|)  (new_scripy.py:417-437); mask None -> plain MSE (MNIST_script.py:252)"""

    @staticmethod
    def forward(ctx, pred, noise, mask, cfg6):
        L.require_device(pred, noise)
        pred, noise = pred.contiguous(), noise.contiguous()
        B, Cc, H, W = pred.shape
        loss = _zeros((1,), torch.float32, pred)
        call("dm_loss_fwd", ptr(pred), ptr(noise), ptr(mask), ptr(cfg6), ptr(loss), B, Cc, H, W)
        ctx.save_for_backward(pred, noise, mask, cfg6)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        pred, noise, mask, cfg6 = ctx.saved_tensors
        B, Cc, H, W = pred.shape
        dpred = _empty(pred.shape, torch.float32, pred)
        gs = g.reshape(1).float().contiguous()
        call("dm_loss_bwd", ptr(pred), ptr(noise), ptr(mask), ptr(cfg6), ptr(gs), ptr(dpred), L.DM_F32, B, Cc, H, W, 0)
        return dpred, None, None, None


def cfg_update(x, eps2n, z, guide_w, sched, step, seed=0, dec_step=True, first_elem=0):
    """In-place x_{t-1} update (new_scripy.py:468-475); `step` is a device int32 tensor holding i.  `first_elem`: x is the
    slice starting at that element of a larger batch (sharded sampling draws the matching slice of the noise stream)."""
    call("dm_cfg_update_slice", ptr(x), ptr(eps2n), ptr(z), float(guide_w), ptr(sched["oneover_sqrta"]), ptr(sched["mab_over_sqrtmab"]),
         ptr(sched["sqrt_beta_t"]), ptr(step), int(seed), x.numel(), int(first_elem), int(dec_step))


def fill_t(t, step, n_T):
    call("dm_fill_t", ptr(t), ptr(step), int(n_T), t.numel())


def randn(shape, device, seed, offset, first_elem=0):
    """Philox N(0,1); `offset` is a Python int, or a device int64 tensor that is read when the kernel runs.
    `first_elem` (multiple of 4): the result is elements [first_elem, first_elem + numel) of the stream."""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    if isinstance(offset, torch.Tensor):
        if first_elem:
            raise L.DmError("randn: a device-resident offset cannot be combined with first_elem")
        call("dm_randn_dev", ptr(out), out.numel(), int(seed), ptr(offset))
    else:
        call("dm_randn_slice", ptr(out), out.numel(), int(seed), int(offset), int(first_elem))
    return out


class MaskAxpy(torch.autograd.Function):
    """out = x + y * [mask > thresh]  (LocalEnhancer, new_scripy.py:172-174); mask (B,H,W) fp32"""

    @staticmethod
    def forward(ctx, x, y, mask, thresh):
        x, y = x.contiguous(), y.contiguous()
        B, H, W, Cc = x.shape
        out = _empty(x.shape, x.dtype, x)
        call("dm_mask_axpy", ptr(x), ptr(y), ptr(mask), thresh, ptr(out), dt(x), B * H * W, Cc)
        ctx.save_for_backward(mask)
        ctx.thresh = thresh
        return out

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        g = g.contiguous()
        B, H, W, Cc = g.shape
        dy = _empty(g.shape, g.dtype, g)
        call("dm_mask_axpy", None, ptr(g), ptr(mask), ctx.thresh, ptr(dy), dt(g), B * H * W, Cc)
        return g, dy, None, None


def refuse_second_backward(namespace):
    """Every autograd.Function of the operator layer runs its backward ONCE per forward.  The backward kernels hand gradients between
    nodes outside autograd (GradFork stashes, BatchNorm-slot and small-gradient accumulators in the zero arena, buffers recycled by
    the optimiser step), so a second pass over the same graph (`retain_graph=True`, `torch.autograd.grad` after `backward`) would
    accumulate into state the first pass already consumed — measured: per-parameter gradients off by up to 60 % from 2 x one pass,
    with no error.  The reference's train loop never does this (new_scripy.py:786-803: one backward per forward); here it raises."""
    for obj in list(namespace.values()):
        if not (isinstance(obj, type) and issubclass(obj, torch.autograd.Function) and obj is not torch.autograd.Function):
            continue
        if "backward" not in obj.__dict__ or obj.__dict__.get("_dm_once", False):
            continue
        inner = obj.__dict__["backward"]
        inner = inner.__func__ if isinstance(inner, staticmethod) else inner

        def make(inner, name):
            def backward(ctx, *grads):
                if getattr(ctx, "_dm_ran", False):
                    raise L.DmError(f"{name}: a second backward pass over the same graph is not supported on the HIP path (gradients "
                                    "travel between nodes outside autograd; run the forward again)")
                ctx._dm_ran = True
                return inner(ctx, *grads)
            return backward

        obj.backward = staticmethod(make(inner, obj.__name__))
        obj._dm_once = True


refuse_second_backward(globals())
