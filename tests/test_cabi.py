"""The C-ABI boundary: every symbol include/dm_amd.h declares is exported by libdm_amd.so, and the
ctypes prototypes in diffusionmodel_amd/_lib.py agree with the header (CPU only, no compute calls)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dm_amd.h")


def header_decls():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(dm_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(3).replace("\n", " ").split(",")]
        if args == ["void"]:
            args = []
        decls[m.group(2)] = args
    return decls


def ctype_of(arg):
    if "dm_stream_t" in arg:
        return "stream"
    if "*" in arg:
        return "ptr"
    if "int64_t" in arg and "uint64_t" not in arg:
        return C.c_int64
    if "uint64_t" in arg:
        return C.c_uint64
    if "uint32_t" in arg:
        return C.c_uint32
    if arg.startswith("float"):
        return C.c_float
    if arg.startswith("int"):
        return C.c_int32
    raise AssertionError("unparsed arg: " + arg)


@pytest.fixture(scope="module")
def lib():
    from diffusionmodel_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib


def test_every_declared_symbol_is_exported(lib):
    decls = header_decls()
    assert len(decls) >= 50
    handle = C.CDLL(lib.LIB_PATH)
    for name in decls:
        assert hasattr(handle, name), f"{name} declared in dm_amd.h but not exported"
    assert set(lib.EXPORTED) == set(decls), set(lib.EXPORTED) ^ set(decls)


def test_ctypes_prototypes_match_header(lib):
    decls = header_decls()
    for name, proto in lib._PROTOS.items():
        args = decls[name]
        assert "dm_stream_t" in args[-1], name
        want = [ctype_of(a) for a in args[:-1]]
        assert len(want) == len(proto), (name, len(want), len(proto))
        for i, (w, p) in enumerate(zip(want, proto)):
            if w == "ptr":
                assert p in (C.c_void_p, C.c_char_p) or issubclass(p, C._Pointer), (name, i, args[i])
            else:
                assert p is w, (name, i, args[i], p)


def test_struct_layouts_match_header(lib):
    src = open(HEADER).read()
    for sname, cls in (("DmConv", lib.DmConv), ("DmWgrad", lib.DmWgrad)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (sname, sname), src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            is_ptr = "*" in stmt
            stmt = re.sub(r"^(const\s+)?(void|float|int32_t)\s*\*?", "", stmt).strip()
            for n in stmt.split(","):
                names.append((n.replace("*", "").strip(), is_ptr))
        assert [n for n, _ in names] == [f[0] for f in cls._fields_], sname
        for (n, is_ptr), f in zip(names, cls._fields_):
            assert (f[1] is C.c_void_p) == is_ptr, (sname, n)


def test_load_and_version(lib):
    h = lib.load()
    assert h.dm_version() >= 100
    assert lib.colstat_blocks(1000) == 125 and lib.colstat_blocks(1 << 18) == 1024       # 8 .. 256 rows per workgroup, ~1024 workgroups


def test_product_refuses_cpu_tensors(lib):
    import torch
    with pytest.raises(lib.DmError):
        lib.require_device(torch.zeros(1))


def test_collective_entry_points_fail_loudly_without_rccl(lib):
    """dm_allreduce_bucket(ptr, n, dtype, comm, stream) is part of the C ABI (include/dm_amd.h, comm.hip binds librccl.so
    at run time).  No GPU here: a bad library path is refused with DM_EUNSUPPORTED and a message, a null communicator with DM_EINVAL."""
    h = lib.load()
    assert h.dm_comm_load(b"/nonexistent/librccl.so") == -2 and b"librccl" in h.dm_last_error()
    assert h.dm_allreduce_bucket(C.c_void_p(16), 4, 0, None, None) == -1
    assert h.dm_comm_destroy(None) == 0


@pytest.mark.gpu
def test_c_abi_allreduce_bucket_world_1_over_rccl():
    """The collective of the C ABI on one rank (a one-GPU box): unique id -> communicator -> in-place SUM all-reduce of fp32 and bf16
    buckets on torch's current stream (with one rank the sum is the input) -> destroy.  N > 1 needs one GPU per rank (RCCL refuses two
    ranks on a device); the multi-rank arithmetic is covered through torch.distributed (tests/test_host_logic.py, gloo)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from diffusionmodel_amd.parallel import CAbiComm
    comm = CAbiComm(rank=0, world=1)
    g = torch.randn(1 << 20, device="cuda:0")
    want = g.clone()
    comm.all_reduce(g)
    comm.all_reduce(g[1024:4096])
    h = torch.randn(4096, device="cuda:0").bfloat16()
    hw = h.clone()
    comm.all_reduce(h)
    torch.cuda.synchronize()
    assert torch.equal(g, want) and torch.equal(h, hw)
    comm.close()
