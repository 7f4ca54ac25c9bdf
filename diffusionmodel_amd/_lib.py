"""ctypes binding of libdm_amd.so (include/dm_amd.h).

The product path has NO fallback: if the shared library is missing or a call fails, an exception is
raised.  Tensors cross the boundary as raw device pointers (`tensor.data_ptr()`), sizes as plain
ints, and every launch goes to torch's current HIP stream, so calls are capturable in a hipGraph.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DM_LIB_PATH") or os.path.join(_HERE, "libdm_amd.so")     # (DM_LIB_PATH: experimental builds)
DEFAULT_CONV_VARIANT = 5      # what libdm_amd.so starts with (igemm.hip g_variant); DM_CONV_VARIANT overrides

DM_F32, DM_BF16, DM_F16 = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3

vp, i32, i64, u64, f32, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_uint32


class DmConv(C.Structure):
    _fields_ = [(n, vp) for n in ("in1", "in2", "w", "scale", "shift", "out", "psum", "psq")] + [
        (n, i32) for n in ("dtype", "act", "out_nchw_f32", "B", "Hi", "Wi", "C1", "C2", "Hq", "Wq", "sy", "sx", "T", "KW",
                           "ty", "tx", "oy0", "ox0", "Ho", "Wo", "osy", "osx", "ooy", "oox", "N", "ldw", "ldc", "coff", "in2_batch",
                           "stat_slots")] + [("addend", vp)]


class DmWgrad(C.Structure):
    _fields_ = [(n, vp) for n in ("dy", "in1", "in2", "dw", "dbias")] + [
        (n, i32) for n in ("dtype", "B", "Hi", "Wi", "C1", "C2", "Hq", "Wq", "sy", "sx", "T", "KW", "ty", "tx", "oy0", "ox0",
                           "Ho", "Wo", "osy", "osx", "ooy", "oox", "N", "ldy", "ldw", "splitk", "overwrite")]


class DmCaChain(C.Structure):
    """include/dm_amd.h: the CoordAttn strip chain (chain.hip)."""
    _PTRS1 = ("xh", "xw", "w1h", "b1h", "w1w", "b1w", "bn_h_g", "bn_h_b", "bn_w_g", "bn_w_b", "rm_h", "rv_h", "rm_w", "rv_w", "whw", "bhw", "wwh",
              "bwh", "gam_h", "gam_w", "wch", "bch", "wcw", "bcw", "zh", "zw", "mean_h", "rstd_h", "mean_w", "rstd_w", "ah", "aw", "xhp", "xwp", "lh",
              "lw", "stat", "dlh", "dlw", "gh", "gw", "bnpart", "dh2w", "dw2h", "dxh", "dxw", "d_w1h", "d_b1h", "d_w1w", "d_b1w", "d_bn_h_g",
              "d_bn_h_b", "d_bn_w_g", "d_bn_w_b", "d_whw", "d_bhw", "d_wwh", "d_bwh", "d_gam", "d_wch", "d_bch", "d_wcw", "d_bcw")
    _fields_ = [(n, i32) for n in ("B", "H", "W", "C", "R", "train", "save")] + [("eps", f32), ("momentum", f32)] + [(n, vp) for n in _PTRS1]


# name -> argument ctypes (the trailing dm_stream_t is appended automatically unless noted)
_PROTOS = {
    "dm_conv": [C.POINTER(DmConv)],
    "dm_conv_parity4": [C.POINTER(DmConv)],
    "dm_conv_wgrad": [C.POINTER(DmWgrad)],
    "dm_pack_w": [vp, vp, i32, i32, i32, i32, i32],
    "dm_pack_wT": [vp, vp, i32, i32, i32, i32, i32, C.POINTER(i32), i32],
    "dm_unpad_dw": [vp, vp, i32, i32, i32, i32, i32],
    "dm_col_stats": [vp, i32, i32, i32, vp, vp],
    "dm_bn_finalize": [vp, vp, i32, i32, i32, f32, f32, vp, vp, vp, vp],
    "dm_bn_act_fwd": [vp, vp, i32, i32, i32, vp, vp, vp, vp, i32],
    "dm_bn_act_bwd_reduce": [vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp],
    "dm_col_reduce": [vp, i32, i32, vp, i32],
    "dm_col_reduce2": [vp, vp, i32, i32, vp, vp],
    "dm_bn_act_bwd_apply": [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp],
    "dm_bn_fold": [vp, vp, vp, vp, vp, f32, i32, vp, vp],
    "dm_bn_act_fwd_slots": [vp, vp, i32, i32, i32, vp, vp, i32, f32, f32, vp, vp, i32, vp, vp, vp, vp],
    "dm_bn_act_bwd_reduce_slots": [vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, i32],
    "dm_bn_act_bwd_apply_slots": [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, i32, vp, vp],
    "dm_gn_act_fwd": [vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, i32, vp, vp],
    "dm_gn_act_bwd": [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp],
    "dm_pool_hw": [vp, i32, i32, i32, i32, vp],
    "dm_scale_residual_fwd": [vp, vp, vp, vp, i32, i32, i32, i32, f32],
    "dm_scale_residual_bwd_reduce": [vp, vp, i32, i32, i32, i32, f32, vp],
    "dm_scale_residual_bwd_apply": [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32],
    "dm_se_fwd": [vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp],
    "dm_se_bwd": [vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp],
    "dm_ca_chain_fwd": [C.POINTER(DmCaChain)],
    "dm_ca_chain_bwd": [C.POINTER(DmCaChain)],
    "dm_ca_pool_fwd": [vp, i32, i32, i32, i32, i32, vp, vp],
    "dm_ca_pool_bwd": [vp, vp, vp, vp, i32, i32, i32, i32, i32],
    "dm_ca_gate_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32],
    "dm_ca_gate_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32],
    "dm_sigmix_fwd": [vp, vp, vp, vp, i32],
    "dm_sigmix_bwd": [vp, vp, vp, vp, vp, i32],
    "dm_linear_fwd": [vp, vp, vp, vp, i32, i32, i32, i32],
    "dm_linear_bwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32],
    "dm_act_fwd": [vp, vp, i32, i32],
    "dm_act_bwd": [vp, vp, vp, i32, i32],
    "dm_onehot_mask": [vp, vp, vp, i32, i32, i32],
    "dm_nchw_to_nhwc": [vp, vp, i32, i32, i32, i32, i32, i32, i32],
    "dm_nhwc_to_nchw": [vp, vp, i32, i32, i32, i32, i32, i32],
    "dm_cast": [vp, vp, i32, i32, i64],
    "dm_film_fwd": [vp, vp, vp, vp, i32, i32, i32, i32],
    "dm_film_bwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32],
    "dm_upcat_fwd": [vp, vp, vp, i32, i32, i32, i32, i32, i32],
    "dm_upcat_fwd_bcast": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32],
    "dm_upcat_bwd": [vp, vp, vp, i32, i32, i32, i32, i32, i32],
    "dm_cat_fwd": [vp, vp, vp, i32, i32, i32, i32],
    "dm_cat_bwd": [vp, vp, vp, i32, i32, i32, i32],
    "dm_avgpool_gelu_fwd": [vp, vp, i32, i32, i32, i32, i32, i32],
    "dm_avgpool_gelu_bwd": [vp, vp, vp, i32, i32, i32, i32, i32, i32],
    "dm_maxpool2_fwd": [vp, vp, i32, i32, i32, i32, i32],
    "dm_maxpool2_bwd": [vp, vp, vp, i32, i32, i32, i32, i32],
    "dm_add": [vp, vp, vp, i32, i64],
    "dm_mask_axpy": [vp, vp, vp, f32, vp, i32, i64, i32],
    "dm_scatter_copy": [vp, i32, i32],
    "dm_zero_ranges": [vp, i32],
    "dm_image_moments": [vp, vp, vp, i32, i64],
    "dm_attn_mask": [vp, vp, i32, i32, f32, f32, f32],
    "dm_draw_ts_keep": [vp, vp, vp, i32, i32, f32, u64, vp],
    "dm_qsample": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32],
    "dm_loss_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32],
    "dm_loss_bwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32],
    "dm_cfg_update": [vp, vp, vp, f32, vp, vp, vp, vp, u64, i64, i32],
    "dm_cfg_update_slice": [vp, vp, vp, f32, vp, vp, vp, vp, u64, i64, i64, i32],
    "dm_fill_t": [vp, vp, i32, i32],
    "dm_randn": [vp, i64, u64, u64],
    "dm_randn_slice": [vp, i64, u64, u64, i64],
    "dm_sumsq": [vp, i64, vp],
    "dm_adamw": [vp, vp, vp, vp, i64, vp, vp, vp, vp],
    "dm_adamw_scaled": [vp, vp, vp, vp, i64, vp, vp, vp, i32, vp, vp],
    "dm_scaler_update": [vp, vp, vp, f32, f32, i32],
    "dm_randn_dev": [vp, i64, u64, vp],
    "dm_pack_multi": [vp, vp, vp, i32],
    "dm_allreduce_bucket": [vp, i64, i32, vp],
    "dm_debug_poison_lds": [u32],
    "dm_plan_marker": [i32],
    "dm_plan_run": [vp, i32, i32],
    "dm_plan_run_timed": [vp, i32, i32, C.c_char_p],
    "dm_plan_timed_results": [vp, vp, vp, i32, vp],
}
_NO_STREAM = {"dm_last_conv_path": ([], i32), "dm_last_wgrad_path": ([], i32), "dm_get_conv_variant": ([], i32), "dm_set_conv_tap4": ([i32], i32), "dm_set_conv_persist": ([i32], i32), "dm_set_conv_packtap": ([i32], i32), "dm_last_conv_persistent": ([], i32), "dm_set_wgrad_pw": ([i32, i32, i32], i32), "dm_set_wgrad_tap4": ([i32], i32), "dm_set_wgrad_skinny": ([i32], i32), "dm_set_conv_wave4": ([i32], i32), "dm_set_splitk_inkernel": ([i32], i32), "dm_set_workspace": ([vp, i64], i32), "dm_set_conv_variant": ([i32], i32), "dm_set_wgrad_variant": ([i32], i32), "dm_version": ([], i32), "dm_last_error": ([], C.c_char_p), "dm_colstat_blocks": ([i32], i32),
              "dm_plan_from_graph": ([vp, C.POINTER(vp)], i32), "dm_plan_info": ([vp, C.POINTER(i32)], i32),
              "dm_plan_segment_marker": ([vp, i32], i32), "dm_plan_op_name": ([vp, i32, C.c_char_p, i32], i32),
              "dm_plan_destroy": ([vp], i32),
              "dm_comm_load": ([C.c_char_p], i32), "dm_comm_unique_id": ([vp], i32), "dm_comm_init": ([C.POINTER(vp), i32, i32, vp], i32),
              "dm_comm_destroy": ([vp], i32)}

EXPORTED = sorted(list(_PROTOS) + list(_NO_STREAM))

_lib = None


class DmError(RuntimeError):
    pass


def load():
    """Load libdm_amd.so (once).  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise DmError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(or `make -C diffusionmodel_amd/csrc`). diffusionmodel_amd has no CPU/ATen fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, args in _PROTOS.items():
        fn = getattr(lib, name)
        fn.argtypes = list(args) + [vp]
        fn.restype = i32
    for name, (args, res) in _NO_STREAM.items():
        fn = getattr(lib, name)
        fn.argtypes = list(args)
        fn.restype = res
    _lib = lib
    v = os.environ.get("DM_WGRAD_VARIANT")
    if v and lib.dm_set_wgrad_variant(int(v)) != 0:
        raise DmError(lib.dm_last_error().decode())
    v = os.environ.get("DM_CONV_TAP4")         # 0: the 4x4 / stride-2 layers stay on the gather kernel (A/B measurements)
    if v is not None:
        lib.dm_set_conv_tap4(int(v))
    v = os.environ.get("DM_CONV_PERSIST")      # 0: one workgroup per tile on every 3x3 launch (A/B measurements)
    if v is not None:
        lib.dm_set_conv_persist(int(v))
    v = os.environ.get("DM_CONV_PACKTAP")      # 0: 8-channel inputs of 3x3 layers stay on the halo kernel's partial chunk (A/B measurements)
    if v is not None:
        lib.dm_set_conv_packtap(int(v))
    v = os.environ.get("DM_WGRAD_SKINNY")      # 0: the stem / head weight gradients stay on the halo kernel (A/B measurements)
    if v is not None:
        lib.dm_set_wgrad_skinny(int(v))
    v = os.environ.get("DM_CONV_WAVE4")        # 1 / 2: the four-wave form of the 3x3 halo kernel (r04 experiment, igemm_halo4.hip)
    if v is not None and lib.dm_set_conv_wave4(int(v)) != 0:
        raise DmError(lib.dm_last_error().decode())
    v = os.environ.get("DM_WGRAD_TAP4")        # 0: the 4x4 / stride-2 weight gradients stay on the per-tap kernel
    if v is not None:
        lib.dm_set_wgrad_tap4(int(v))
    v = os.environ.get("DM_WGRAD_PW")          # 0: the 1x1 weight gradients stay on the per-tap kernel; N > 1: workgroup target of the pixel split
    if v is not None:
        lib.dm_set_wgrad_pw(1 if int(v) else 0, int(v) if int(v) > 1 else 0, -1)
    v = os.environ.get("DM_SPLITK_INKERNEL")   # 0: separate split-K epilogue launches (A/B measurements)
    if v is not None:
        lib.dm_set_splitk_inkernel(int(v))
    v = os.environ.get("DM_CONV_VARIANT")      # tuning knob, see dm_set_conv_variant in dm_amd.h
    if v:
        if lib.dm_set_conv_variant(int(v)) != 0:
            raise DmError(lib.dm_last_error().decode())
    return lib


_workspace = None
WORKSPACE_BYTES = 160 << 20      # split partial sums of the weight-gradient kernels (dm_set_workspace)


def ensure_workspace():
    """Allocate the library's scratch buffer on the current device once and register it (dm_set_workspace)."""
    global _workspace
    if _workspace is None:
        lib = load()
        _workspace = torch.empty(WORKSPACE_BYTES // 4, dtype=torch.float32, device="cuda")
        if lib.dm_set_workspace(_workspace.data_ptr(), WORKSPACE_BYTES) != 0:
            raise DmError(lib.dm_last_error().decode())
        device_guard()                 # first MFMA launch of this process on the device: is another process of ours on it?
    return _workspace


# ---- one process per GPU, enforced per PROCESS (not only per torch.distributed job) ----------------------------------------
# Workgroups that hold more than 64 KiB of LDS do not survive being preempted for ANOTHER PROCESS on this driver stack
# (the compute-wave save area restores the first 64 KiB of LDS only — scripts/probes/lds_preempt.hip).
# Every process that loads this library therefore holds a shared flock on a per-device lock file for its lifetime; a process
# that finds (or later gets) company on its device switches itself to the <= 64-KiB kernel variants and says so on stderr.
_guard = {"fd": None, "path": None, "shared": False, "checked": 0}
LOCK_DIR = os.environ.get("DM_LOCK_DIR", "/tmp")


def device_identity(index=None):
    """A string that is the same for every process using the same physical GPU on this host (uuid / PCI address), else the
    visible-device string + index."""
    vis = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "")))
    ident = None
    if torch.cuda.is_available():
        index = torch.cuda.current_device() if index is None else index
        if index < torch.cuda.device_count():
            props = torch.cuda.get_device_properties(index)
            ident = str(getattr(props, "uuid", "")) or None
            bus = getattr(props, "pci_bus_id", None)
            if bus is not None:
                ident = f"{ident}|bus{bus}|dev{getattr(props, 'pci_device_id', '')}|dom{getattr(props, 'pci_domain_id', '')}"
    if ident is None:
        ident = f"visible[{vis}]#{0 if index is None else index}"
    return ident


def _select_small_lds(reason):
    import sys
    lib = load()
    if lib.dm_set_conv_variant(2) != 0 or lib.dm_set_wgrad_variant(2) != 0:
        raise DmError(lib.dm_last_error().decode())
    _guard["shared"] = True
    print(f"[diffusionmodel_amd] pid {os.getpid()}: {reason}: selecting the <= 64-KiB-LDS kernel variants (DM_CONV_VARIANT=2, "
          "DM_WGRAD_VARIANT=2); workgroups with more LDS are not preemption-safe between processes on this stack. "
          "Use one process per GPU for full speed.", file=sys.stderr, flush=True)


def device_guard(identity=None, recheck=False):
    """Take (first call) or re-test (recheck=True) this process's claim on its device.  Returns True when the device is shared with
    another process of this library — the library is then on its <= 64-KiB-LDS kernels.  DM_DEVICE_GUARD=0 switches the guard off
    (several processes on one GPU with the full-size kernels: at your own risk)."""
    import fcntl
    if os.environ.get("DM_DEVICE_GUARD", "1") == "0":
        return False
    if _guard["shared"]:
        return True
    if _guard["fd"] is None:
        import hashlib
        ident = device_identity() if identity is None else identity
        path = os.path.join(LOCK_DIR, "diffusionmodel_amd.gpu-" + hashlib.sha1(str(ident).encode()).hexdigest()[:16] + ".lock")
        try:
            fd = os.open(path, os.O_CREAT | os.O_RDWR, 0o666)
        except OSError:
            return False                                   # no writable lock directory: nothing to go by
        _guard["fd"], _guard["path"] = fd, path
        try:
            fcntl.flock(fd, fcntl.LOCK_SH | fcntl.LOCK_NB)
        except OSError:                                    # somebody is probing with an exclusive lock right now: we are not alone
            fcntl.flock(fd, fcntl.LOCK_SH)
            _select_small_lds("another process is using this GPU")
            return True
    fd = _guard["fd"]
    _guard["checked"] += 1
    try:                                                   # sole holder <=> the shared lock can be made exclusive
        fcntl.flock(fd, fcntl.LOCK_EX | fcntl.LOCK_NB)
        fcntl.flock(fd, fcntl.LOCK_SH)
        return False
    except OSError:
        fcntl.flock(fd, fcntl.LOCK_SH)                     # (a failed conversion may drop the lock: take it again)
        _select_small_lds("another process " + ("started using" if recheck else "is using") + " this GPU")
        return True


def device_is_shared():
    return _guard["shared"]


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", torch.cuda.current_device)


def _stream():
    """Raw handle of torch's current HIP stream on the current device.  torch.cuda.current_stream() costs ~9 us of Python per
    call (1200 launches per train step); the private raw-stream getter is a plain C call."""
    if _raw_stream is not None:
        return _raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def dt(t_or_dtype):
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    if d == torch.float32:
        return DM_F32
    if d == torch.bfloat16:
        return DM_BF16
    if d == torch.float16:
        return DM_F16
    raise DmError(f"unsupported dtype {d}: the HIP path computes in float32, bfloat16 or float16")


POISON_CALLS = [0]
POISON_LDS = [None]       # tests: a 32-bit pattern -> every library call is preceded by dm_debug_poison_lds(pattern) on the same stream


def call(name, *args):
    """Invoke an entry point on the current stream; raise DmError on failure."""
    lib = load()
    if POISON_LDS[0] is not None and name != "dm_debug_poison_lds":
        POISON_CALLS[0] += 1
        if lib.dm_debug_poison_lds(int(POISON_LDS[0]), _stream()) != 0:
            raise DmError("dm_debug_poison_lds: " + lib.dm_last_error().decode())
    rc = getattr(lib, name)(*args, _stream())
    if rc != 0:
        raise DmError(f"{name} failed (rc={rc}): {lib.dm_last_error().decode()}")


def colstat_blocks(m):
    return load().dm_colstat_blocks(int(m))


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DmError("diffusionmodel_amd runs on a HIP device only (got a CPU tensor). "
                          "There is deliberately no CPU fallback; the CPU restatement lives in oracle/ for tests.")
