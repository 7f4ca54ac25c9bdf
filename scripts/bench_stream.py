"""Micro-benchmark of the HBM-bound streaming kernels (GPU): achieved GB/s against algorithmic bytes.

    python scripts/bench_stream.py [--dtype bf16|fp32]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    from diffusionmodel_amd import _lib as L
    from diffusionmodel_amd.ops import call, ptr
    dev = "cuda:0"
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    es = 2 if a.dtype == "bf16" else 4
    d = L.dt(dtype)
    for (B, H, C) in [(64, 64, 128), (64, 32, 256), (64, 16, 512), (64, 8, 1024), (64, 64, 32)]:
        M = B * H * H
        z = torch.randn(M, C, device=dev).to(dtype)
        dy = torch.randn(M, C, device=dev).to(dtype)
        out = torch.empty_like(z)
        mean, rstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        nblk = L.colstat_blocks(M)
        p1, p2 = torch.empty(nblk, C, device=dev), torch.empty(nblk, C, device=dev)
        s1, s2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        nbytes = M * C * es
        t = timeit(lambda: call("dm_bn_act_fwd", ptr(z), ptr(out), d, M, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), L.ACT_GELU))
        line = f"[{B}x{H}x{H}x{C}] {nbytes / 1e6:6.1f} MB  bn_act_fwd {t * 1e6:7.1f} us {2 * nbytes / t / 1e9:7.0f} GB/s"
        t = timeit(lambda: call("dm_bn_act_bwd_reduce", ptr(z), ptr(dy), d, M, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), L.ACT_GELU, ptr(p1), ptr(p2)))
        line += f" | bwd_reduce {t * 1e6:7.1f} us {2 * nbytes / t / 1e9:7.0f} GB/s"
        t = timeit(lambda: call("dm_bn_act_bwd_apply", ptr(z), ptr(dy), ptr(out), d, M, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), L.ACT_GELU, ptr(s1), ptr(s2)))
        line += f" | bwd_apply {t * 1e6:7.1f} us {3 * nbytes / t / 1e9:7.0f} GB/s"
        t = timeit(lambda: call("dm_col_stats", ptr(z), d, M, C, ptr(p1), ptr(p2)))
        line += f" | col_stats {t * 1e6:7.1f} us {nbytes / t / 1e9:7.0f} GB/s"
        t = timeit(lambda: call("dm_cast", ptr(z), ptr(out), d, d, M * C))
        line += f" | copy {t * 1e6:7.1f} us {2 * nbytes / t / 1e9:7.0f} GB/s"
        print(line)


if __name__ == "__main__":
    main()
