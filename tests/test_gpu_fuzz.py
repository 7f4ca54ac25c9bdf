"""Seeded random sweep over convolution shapes: every dispatch boundary of dm_conv / dm_conv_wgrad (halo, four-tap, pointwise and
gather kernels; 1x1 weight-gradient kernel vs per-tap kernel) is crossed by some case.  Small-integer data, so forward, input
gradient, weight gradient and bias gradient must equal torch's fp32 result exactly."""
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cases():
    rnd = random.Random(20260)
    out = []
    chans = [8, 16, 24, 32, 40, 64, 72, 96, 128, 136, 160, 192, 256, 384]
    for i in range(48):
        k, s = rnd.choice([(1, 1), (1, 1), (3, 1), (3, 1), (4, 2)])
        H = rnd.choice([4, 6, 8, 10, 12, 16, 20, 32] if k != 4 else [4, 8, 12, 16, 32])
        W = H if rnd.random() < 0.7 else rnd.choice([4, 8, 12, 16, 24])
        if k == 4:
            W = H
        B = rnd.choice([1, 2, 3, 5])
        C, N = rnd.choice(chans), rnd.choice(chans)
        out.append((k, s, B, C, N, H, W))
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_random_conv_shapes_exact_integers(dtype):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from diffusionmodel_amd import ops as o, _lib
    lib = _lib.load()
    paths = {}
    for (k, s, B, C, N, H, W) in _cases():
        g = torch.Generator().manual_seed(k * 7 + B * 1000 + C * 31 + N + H)
        ri = lambda *sh: torch.randint(-2, 3, sh, generator=g).float()
        dens = min(1.0, 24.0 / (C * k * k))                                       # keep |y| < 256: exact in bf16
        pad = 1 if k == 4 else k // 2
        x, b, probe = ri(B, C, H, W), ri(N), ri(B, N, (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1)
        w = ri(N, C, k, k) * (torch.rand(N, C, k, k, generator=g) < dens).float()
        probe = probe * (torch.rand(probe.shape, generator=g) < min(1.0, 24.0 / (N * (k // s) ** 2))).float()
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = F.conv2d(xr, wr, br, stride=s, padding=pad)
        assert yr.shape == probe.shape
        (yr * probe).sum().backward()
        if yr.abs().max() >= 256 or xr.grad.abs().max() >= 256:
            continue
        wd = torch.nn.Parameter(w.to(DEV).contiguous(memory_format=torch.channels_last))
        bd = torch.nn.Parameter(b.to(DEV))
        holder = type("H", (), {"weight": wd, "bias": bd})()
        xd = x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype).requires_grad_(True)
        y = o.conv_bn_act(xd, None, holder, None, o.ConvSpec(k, k, s, pad))
        fp = lib.dm_last_conv_path()
        tag = (k, s, B, C, N, H, W)
        assert torch.equal(y.float().cpu().permute(0, 3, 1, 2), yr.detach()), ("forward", tag, fp)
        (y.float() * probe.permute(0, 2, 3, 1).contiguous().to(DEV)).sum().backward()
        wp = lib.dm_last_wgrad_path()
        paths[(k, fp, wp)] = paths.get((k, fp, wp), 0) + 1
        assert torch.equal(xd.grad.float().cpu().permute(0, 3, 1, 2), xr.grad), ("input gradient", tag, fp)
        assert torch.equal(wd.grad.cpu(), wr.grad), ("weight gradient", tag, wp)
        assert torch.equal(bd.grad.cpu(), br.grad), ("bias gradient", tag, wp)
    print("kernel paths taken (kernel size, forward path, wgrad path): count", paths)
    assert len(paths) >= 6, paths          # the sweep really crosses the dispatch boundaries
