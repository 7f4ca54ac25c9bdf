"""Repeat each MFMA kernel on the SAME operands and compare every result with the first one bit for bit (the halo convolution with
plain epilogue, its flipped input-gradient form, the halo weight gradient + reduce are deterministic by construction): a staging
race — an LDS-DMA piece landing in a buffer another wave still reads, a wait that does not cover a piece — shows up as a mismatch.
A different layer is launched between repetitions so that caches and timing move.   python scripts/race_repeat.py [reps]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusionmodel_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = "cuda:0"
SHAPES = [("64^2 128->128", 64, 64, 128, 128), ("64^2 256->128", 64, 64, 256, 128), ("32^2 256->256", 64, 32, 256, 256),
          ("16^2 512->512", 64, 16, 512, 512), ("8^2 1024->1024", 64, 8, 1024, 1024), ("8^2 2048->512", 64, 8, 2048, 512),
          ("128^2 256->256", 4, 128, 256, 256), ("64^2 8->128 (stem, packed-tap)", 64, 64, 8, 128), ("64^2 128->8 (head, narrow)", 64, 64, 128, 8)]
bad = 0
for dtype in (torch.bfloat16, torch.float16):
    cases = []
    for name, B, H, Ci, Co in SHAPES:
        g = torch.Generator(device=dev).manual_seed(hash(name) & 0xffff)
        x = torch.randn(B, H, H, Ci, device=dev, generator=g).to(dtype)
        w = (torch.randn(Co, 3, 3, Ci, device=dev, generator=g) / (Ci * 9) ** 0.5).to(dtype)
        dy = torch.randn(B, H, H, Co, device=dev, generator=g).to(dtype)
        geom = dict(dtype=dtype, B=B, Hi=H, Wi=H, C1=Ci, C2=0, Hq=H, Wq=H, sy=1, sx=1, T=9, KW=3, ty=1, tx=1, oy0=-1, ox0=-1, Ho=H, Wo=H, N=Co)
        cases.append((name, x, w, dy, geom, B, H, Ci, Co))
    def fwd(c):
        name, x, w, dy, geom, B, H, Ci, Co = c
        y = torch.empty(B, H, H, Co, device=dev, dtype=dtype)
        ops._conv_call(x, None, w.data_ptr(), 9 * Ci, y, **geom)
        return y
    def wg(c):
        name, x, w, dy, geom, B, H, Ci, Co = c
        dw = torch.zeros(Co, 3, 3, Ci, device=dev); db = torch.zeros(Co, device=dev)
        ops._wgrad_call(dy, x, None, dw, db, ldy=Co, ldw=9 * Ci, **geom)
        return torch.cat([dw.reshape(-1), db])
    for kind, fn in (("conv", fwd), ("wgrad", wg)):
        for i, c in enumerate(cases):
            if kind == "wgrad" and min(c[7], c[8]) < 64:
                continue                                     # (the skinny weight-gradient kernel's slices meet in dw through atomics)
            ref = fn(c).clone()
            miss = 0
            for r in range(reps):
                fn(cases[(i + 1 + r) % len(cases)])          # another layer in between
                out = fn(c)
                if not torch.equal(out, ref):
                    miss += 1
                    if miss == 1:
                        d = (out.float() - ref.float()).abs()
                        print(f"   first mismatch {kind} {c[0]} {dtype}: {int((d > 0).sum())} elements, max {float(d.max()):.3e}", flush=True)
            bad += miss
            print(f"{kind:6s} {str(dtype)[6:]:9s} {c[0]:16s} {miss} of {reps} repetitions differ", flush=True)
print("TOTAL mismatching repetitions:", bad)
sys.exit(1 if bad else 0)
