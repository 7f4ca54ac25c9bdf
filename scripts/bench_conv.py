"""Per-shape micro-benchmark of the MFMA kernels on the cfg-2 layer shapes (GPU).

    python scripts/bench_conv.py [--variant V] [--dtype bf16|fp32] [--what fwd,wgrad]

Times dm_conv (forward geometry) and dm_conv_wgrad with HIP events on the launch stream, prints TFLOP/s."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [  # (name, B, H, Cin, Cout, k, stride)
    ("64^2 128->128 3x3", 64, 64, 128, 128, 3, 1),
    ("64^2 256->128 3x3", 64, 64, 256, 128, 3, 1),
    ("32^2 256->256 3x3", 64, 32, 256, 256, 3, 1),
    ("32^2 512->128 3x3", 64, 32, 512, 128, 3, 1),
    ("16^2 512->512 3x3", 64, 16, 512, 512, 3, 1),
    ("8^2 1024->1024 3x3", 64, 8, 1024, 1024, 3, 1),
    ("8^2 2048->512 3x3", 64, 8, 2048, 512, 3, 1),
    ("64^2 128->128 4x4s2", 64, 64, 128, 128, 4, 2),
    ("8^2 1024->1024 4x4s2", 64, 8, 1024, 1024, 4, 2),
    ("64^2 128->32 1x1", 64, 64, 128, 32, 1, 1),
    ("64^2 32->128 1x1", 64, 64, 32, 128, 1, 1),
    ("128^2 256->256 3x3 (cfg-5)", 8, 128, 256, 256, 3, 1),
    ("64^2 128->8 3x3 (head)", 64, 64, 128, 8, 3, 1),
    ("64^2 8->128 3x3 (stem)", 64, 64, 8, 128, 3, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--what", default="fwd,wgrad")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--zeros", action="store_true", help="all-zero operands: same cycles, higher sustained clock (DVFS check)")
    ap.add_argument("--only", default="", help="substring filter on the shape name")
    args = ap.parse_args()
    if args.variant:
        os.environ["DM_CONV_VARIANT"] = str(args.variant)
    from diffusionmodel_amd import ops
    dev = "cuda:0"
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    what = args.what.split(",")
    print(f"variant={args.variant or 'default'} dtype={args.dtype}")
    for name, B, H, Ci, Co, k, s in SHAPES:
        if args.only and args.only not in name:
            continue
        x = torch.randn(B, H, H, Ci, device=dev).to(dtype)
        w = (torch.randn(Co, k, k, Ci, device=dev) / (Ci * k * k) ** 0.5).to(dtype)
        if args.zeros:
            x.zero_(), w.zero_()
        p = (k - 1) // 2 if s == 1 else 1
        Ho = (H + 2 * p - k) // s + 1
        y = torch.empty(B, Ho, Ho, Co, device=dev, dtype=dtype)
        geom = dict(dtype=dtype, B=B, Hi=H, Wi=H, C1=Ci, C2=0, Hq=Ho, Wq=Ho, sy=s, sx=s, T=k * k, KW=k, ty=1, tx=1, oy0=-p, ox0=-p,
                    Ho=Ho, Wo=Ho, N=Co)
        flops = 2.0 * B * Ho * Ho * Co * k * k * Ci
        line = f"{name:30s}"
        if "fwd" in what:
            for _ in range(3):
                ops._conv_call(x, None, w.data_ptr(), k * k * Ci, y, **geom)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                ops._conv_call(x, None, w.data_ptr(), k * k * Ci, y, **geom)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / args.iters
            line += f"  fwd {t * 1e6:8.1f} us {flops / t / 1e12:7.1f} TF/s"
        if "wgrad" in what:
            dw = torch.zeros(Co, k, k, Ci, device=dev)
            db = torch.zeros(Co, device=dev)
            for _ in range(3):
                ops._wgrad_call(y, x, None, dw, db, ldy=Co, ldw=k * k * Ci, **geom)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                ops._wgrad_call(y, x, None, dw, db, ldy=Co, ldw=k * k * Ci, **geom)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e-3 / args.iters
            line += f"  wgrad {t * 1e6:8.1f} us {flops / t / 1e12:7.1f} TF/s"
        print(line)


if __name__ == "__main__":
    main()
