"""Kernel-level parity on a real MI355X: every C-ABI family against a plain PyTorch fp32 (CPU) statement
of the same op.  fp32 mode must agree to fp32 rounding; bf16 mode to bf16 rounding (tolerances stated
per test).  Exact-integer data is used where a layout bug could hide behind smooth data."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.manual_seed(0)


def ops():
    from diffusionmodel_amd import ops as o
    return o


def nhwc(x, dtype=torch.float32):
    return x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)


def nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def rel_err(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12)).item()


class Holder:
    def __init__(self, w, b):
        self.weight = torch.nn.Parameter(w.to(DEV).contiguous(memory_format=torch.channels_last) if w.dim() == 4 else w.to(DEV))
        self.bias = torch.nn.Parameter(b.to(DEV)) if b is not None else None


CONV_CASES = [
    # B, Cin, Cout, H, W, k, stride, pad
    (2, 32, 32, 8, 8, 3, 1, 1),
    (3, 16, 64, 5, 7, 3, 1, 1),          # ragged M (105 rows), non-square
    (2, 64, 136, 16, 16, 3, 1, 1),       # N not a multiple of the tile
    (2, 32, 8, 12, 12, 1, 1, 0),         # 1x1, tiny N
    (2, 32, 32, 16, 16, 4, 2, 1),        # 4x4 stride 2
    (1, 256, 128, 4, 4, 3, 1, 1),        # deep / small spatial
    (2, 8, 16, 6, 6, 3, 1, 1),           # C smaller than one k-step
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_forward_backward(case, dtype):
    o = ops()
    B, Ci, Co, H, W, k, s, p = case
    x = torch.randn(B, Ci, H, W)
    w = torch.randn(Co, Ci, k, k) / math.sqrt(Ci * k * k)
    b = torch.randn(Co) * 0.1
    if dtype == torch.bfloat16:      # make inputs exactly representable so only accumulation order differs
        x, w = x.bfloat16().float(), w.bfloat16().float()
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=s, padding=p)
    probe = torch.randn_like(yr)
    if dtype == torch.bfloat16:
        probe = probe.bfloat16().float()
    (yr * probe).sum().backward()

    conv = Holder(w, b)
    spec = o.ConvSpec(k, k, s, p)
    xd = nhwc(x, dtype).requires_grad_(True)
    y = o.conv_bn_act(xd, None, conv, None, spec)
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    assert rel_err(nchw(y), yr.detach()) < tol
    (y.float() * nhwc(probe)).sum().backward()
    gtol = 5e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(nchw(xd.grad), xr.grad) < gtol, "dgrad"
    assert rel_err(conv.weight.grad.cpu(), wr.grad) < gtol, "wgrad"
    assert rel_err(conv.bias.grad.cpu(), br.grad) < gtol, "bias grad"


def test_wgrad_bf16_exact_integers():
    """Small-integer data makes the bf16 weight gradient exact: any mistake in the transposed LDS read
    (ds_read_b64_tr_b16 lane/row mapping) or in the pixel<->k assignment shows up as an O(1) error."""
    o = ops()
    B, Ci, Co, H, W = 2, 16, 24, 6, 5
    x = torch.randint(-3, 4, (B, Ci, H, W)).float()
    g = torch.randint(-2, 3, (B, Co, H, W)).float()
    w = torch.randint(-2, 3, (Co, Ci, 3, 3)).float()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    (F.conv2d(xr, wr, None, padding=1) * g).sum().backward()
    conv = Holder(w, torch.zeros(Co))
    xd = nhwc(x, torch.bfloat16).requires_grad_(True)
    y = o.conv_bn_act(xd, None, conv, None, o.ConvSpec(3, 3, 1, 1))
    (y.float() * nhwc(g)).sum().backward()
    assert torch.equal(conv.weight.grad.cpu(), wr.grad)
    assert torch.equal(nchw(xd.grad), xr.grad)
    assert torch.equal(conv.bias.grad.cpu(), g.sum((0, 2, 3)))


@pytest.fixture(params=[5], ids=["halo"])
def halo_variant(request):
    """Route eligible 3x3 convolutions through the halo-resident kernel (either schedule) for the duration of a test."""
    from diffusionmodel_amd import _lib
    lib = _lib.load()
    assert lib.dm_set_conv_variant(request.param) == 0
    yield
    assert lib.dm_set_conv_variant(int(__import__("os").environ.get("DM_CONV_VARIANT", "0")) or _lib.DEFAULT_CONV_VARIANT) == 0


HALO_CASES = [
    # B, C1, C2, Cout, H (= W)
    (2, 64, 0, 128, 16),     # one chunk, one n-tile, tile = whole image
    (1, 128, 0, 64, 32),     # two chunks, N < tile, 4 tiles per image (interior halo rows come from neighbours)
    (1, 64, 0, 192, 64),     # two n-tiles, the second half empty; 16 tiles per image
    (2, 64, 128, 64, 16),    # concatenated sources with different channel counts
    (3, 192, 0, 136, 32),    # three chunks, ragged N, odd batch
    (4, 64, 0, 128, 8),      # 8x8 images: four whole images side by side in one halo
    (8, 128, 64, 192, 8),    # 8x8, two tiles, concatenated sources, two n-tiles
    (2, 8, 0, 128, 32),      # one partial chunk (the 8-channel stem / head-gradient layers)
    (4, 8, 0, 24, 16),       # partial chunk and partial n tile
    (1, 64, 0, 64, 128),     # 128-pixel rows: column tiles of 64, halo columns come from the neighbouring tile
    (2, 64, 64, 136, 128),   # column tiles, concatenated sources, ragged N
    (4, 256, 0, 128, 8),     # one tile, four chunks: split over two workgroups, the last to arrive folds the partials and finishes
    (2, 512, 0, 256, 16),    # four tiles x four splits (in-kernel split-K epilogue, two n-tiles)
    (8, 384, 128, 192, 8),   # split-K with concatenated sources and a ragged second n-tile
]


@pytest.mark.parametrize("case", HALO_CASES)
def test_conv_halo_kernel_exact_integers(case, halo_variant):
    """Small-integer bf16 data: every product and partial sum is exact, so the halo kernels (chunk-major
    k order, shifted fragment reads, zero-filling DMA) must reproduce F.conv2d bit for bit — forward,
    the input gradient (same kernel, transposed weights, mirrored taps) and the weight gradient."""
    o = ops()
    B, C1, C2, Co, H = case
    g = torch.Generator().manual_seed(C1 + Co + H)
    ri = lambda *s: torch.randint(-1, 2, s, generator=g).float()
    x1, x2 = ri(B, C1, H, H), (ri(B, C2, H, H) if C2 else None)
    w, b, probe = ri(Co, C1 + C2, 3, 3), ri(Co), ri(B, Co, H, H)
    keep = (torch.rand(Co, C1 + C2, 3, 3, generator=g) < 0.25).float()       # sparse weights keep |y| < 256 (exact in bf16)
    w = w * keep
    r1 = x1.clone().requires_grad_(True)
    r2 = x2.clone().requires_grad_(True) if C2 else None
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(torch.cat((r1, r2), 1) if C2 else r1, wr, br, padding=1)
    assert yr.abs().max() < 256
    (yr * probe).sum().backward()
    conv = Holder(w, b)
    d1 = nhwc(x1, torch.bfloat16).requires_grad_(True)
    d2 = nhwc(x2, torch.bfloat16).requires_grad_(True) if C2 else None
    y = o.conv_bn_act(d1, d2, conv, None, o.ConvSpec(3, 3, 1, 1))
    assert torch.equal(nchw(y), yr.detach())
    (y.float() * nhwc(probe)).sum().backward()
    assert torch.equal(nchw(d1.grad), r1.grad)
    if C2:
        assert torch.equal(nchw(d2.grad), r2.grad)
    # weight / bias gradient (halo-resident wgrad kernel + split reduction): integer sums, exact in fp32
    assert torch.equal(conv.weight.grad.cpu(), wr.grad)
    assert torch.equal(conv.bias.grad.cpu(), br.grad)


def test_conv_halo_kernel_bn_statistics(halo_variant):
    """Train-mode BatchNorm behind the halo kernel: the two 128-row statistics partials a 256-pixel tile emits."""
    o = ops()
    torch.manual_seed(0)
    B, Ci, Co, H = 2, 64, 128, 32
    x = torch.randn(B, Ci, H, H).bfloat16().float()
    conv_r = torch.nn.Conv2d(Ci, Co, 3, 1, 1)
    with torch.no_grad():
        conv_r.weight.copy_(conv_r.weight.bfloat16().float())
    bn_r, bn_d = torch.nn.BatchNorm2d(Co), torch.nn.BatchNorm2d(Co).to(DEV)
    with torch.no_grad():
        bn_r.weight.uniform_(0.5, 1.5); bn_r.bias.uniform_(-0.3, 0.3)
    bn_d.load_state_dict(bn_r.state_dict())
    conv_d = Holder(conv_r.weight.detach().clone(), conv_r.bias.detach().clone())
    xr = x.clone().requires_grad_(True)
    yr = F.gelu(bn_r(conv_r(xr)))
    probe = torch.randn_like(yr)
    (yr * probe).sum().backward()
    spec = o.ConvSpec(3, 3, 1, 1, o.ACT_GELU, bn_d)
    xd = nhwc(x, torch.bfloat16).requires_grad_(True)
    y = o.conv_bn_act(xd, None, conv_d, bn_d, spec)
    # bars from profiles/r02_parity.json ("conv_bn_gelu_kernel_bf16": this very case): HIP bf16 max-rel error y 4.5e-3, dx 3.5e-3,
    # dw 2.3e-3, dgamma 3.0e-3, dbeta 2.5e-3 — each at or below torch's own autocast(bfloat16) run of the same modules (4.5e-3,
    # 4.7e-3, 3.8e-3, 3.6e-3, 3.6e-3); the bars are 2x torch-autocast's error
    assert rel_err(nchw(y), yr.detach()) < 9e-3
    assert rel_err(bn_d.running_mean.cpu(), bn_r.running_mean) < 1e-2
    assert rel_err(bn_d.running_var.cpu(), bn_r.running_var) < 1e-2
    (y.float() * nhwc(probe)).sum().backward()
    assert rel_err(nchw(xd.grad), xr.grad) < 9e-3
    assert rel_err(conv_d.weight.grad.cpu(), conv_r.weight.grad) < 8e-3
    assert rel_err(bn_d.weight.grad.cpu(), bn_r.weight.grad) < 8e-3
    assert rel_err(bn_d.bias.grad.cpu(), bn_r.bias.grad) < 8e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_concat_and_stem_pad_and_nchw_out(dtype):
    o = ops()
    B, H = 2, 8
    # two-source (concat) input
    x1, x2 = torch.randn(B, 16, H, H), torch.randn(B, 24, H, H)
    w = torch.randn(32, 40, 3, 3) / 19
    b = torch.randn(32) * 0.1
    if dtype == torch.bfloat16:
        x1, x2, w = x1.bfloat16().float(), x2.bfloat16().float(), w.bfloat16().float()
    r1, r2, wr = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(torch.cat((r1, r2), 1), wr, b, padding=1)
    probe = torch.randn_like(yr)
    (yr * probe).sum().backward()
    conv = Holder(w, b)
    d1, d2 = nhwc(x1, dtype).requires_grad_(True), nhwc(x2, dtype).requires_grad_(True)
    y = o.conv_bn_act(d1, d2, conv, None, o.ConvSpec(3, 3, 1, 1))
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(nchw(y), yr.detach()) < tol
    (y.float() * nhwc(probe)).sum().backward()
    assert rel_err(nchw(d1.grad), r1.grad) < tol and rel_err(nchw(d2.grad), r2.grad) < tol
    assert rel_err(conv.weight.grad.cpu(), wr.grad) < tol
    # stem: 3 real channels padded to 8; head: Cout = 3 written as NCHW fp32
    xs = torch.randn(B, 3, H, H)
    ws, wh = torch.randn(16, 3, 3, 3) / 5, torch.randn(3, 16, 3, 3) / 12
    bs, bh = torch.randn(16) * 0.1, torch.randn(3) * 0.1
    if dtype == torch.bfloat16:
        xs, ws, wh = xs.bfloat16().float(), ws.bfloat16().float(), wh.bfloat16().float()
    wsr, whr = ws.clone().requires_grad_(True), wh.clone().requires_grad_(True)
    hr = F.conv2d(xs, wsr, bs, padding=1)
    yr = F.conv2d(hr, whr, bh, padding=1)
    probe = torch.randn_like(yr)
    (yr * probe).sum().backward()
    stem, head = Holder(ws, bs), Holder(wh, bh)
    x8 = o.nchw_to_nhwc(xs.to(DEV), dtype, 8)
    h = o.conv_bn_act(x8, None, stem, None, o.ConvSpec(3, 3, 1, 1))
    y = o.conv_bn_act(h, None, head, None, o.ConvSpec(3, 3, 1, 1, out_nchw=True))
    assert y.shape == (B, 3, H, H) and y.dtype == torch.float32
    tol2 = 3e-5 if dtype == torch.float32 else 3e-2
    assert rel_err(y.cpu(), yr.detach()) < tol2
    (y * probe.to(DEV)).sum().backward()
    assert rel_err(stem.weight.grad.cpu(), wsr.grad) < tol2
    assert rel_err(head.weight.grad.cpu(), whr.grad) < tol2
    assert rel_err(head.bias.grad.cpu(), probe.sum((0, 2, 3))) < tol2


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_bn_gelu(train, dtype):
    o = ops()
    B, Ci, Co, H = 4, 16, 32, 8
    x = torch.randn(B, Ci, H, H)
    conv_r = torch.nn.Conv2d(Ci, Co, 3, 1, 1)
    bn_r = torch.nn.BatchNorm2d(Co)
    with torch.no_grad():
        bn_r.weight.uniform_(0.5, 1.5); bn_r.bias.uniform_(-0.3, 0.3)
        bn_r.running_mean.uniform_(-0.2, 0.2); bn_r.running_var.uniform_(0.5, 1.5)
    bn_d = torch.nn.BatchNorm2d(Co).to(DEV)
    bn_d.load_state_dict(bn_r.state_dict())
    conv_d = Holder(conv_r.weight.detach().clone(), conv_r.bias.detach().clone())
    bn_r.train(train); bn_d.train(train)
    xr = x.clone().requires_grad_(True)
    yr = F.gelu(bn_r(conv_r(xr)))
    probe = torch.randn_like(yr)
    (yr * probe).sum().backward()
    spec = o.ConvSpec(3, 3, 1, 1, o.ACT_GELU, bn_d)
    xd = nhwc(x, dtype).requires_grad_(True)
    y = o.conv_bn_act(xd, None, conv_d, bn_d, spec)
    tol = 5e-5 if dtype == torch.float32 else 4e-2
    assert rel_err(nchw(y), yr.detach()) < tol
    (y.float() * nhwc(probe)).sum().backward()
    assert rel_err(nchw(xd.grad), xr.grad) < tol * 4
    assert rel_err(conv_d.weight.grad.cpu(), conv_r.weight.grad) < tol * 4
    assert rel_err(bn_d.weight.grad.cpu(), bn_r.weight.grad) < tol * 4
    assert rel_err(bn_d.bias.grad.cpu(), bn_r.bias.grad) < tol * 4
    if train:
        assert rel_err(bn_d.running_mean.cpu(), bn_r.running_mean) < 1e-5 + (0 if dtype == torch.float32 else 1e-2)
        assert rel_err(bn_d.running_var.cpu(), bn_r.running_var) < 1e-5 + (0 if dtype == torch.float32 else 1e-2)
    # eval-mode fused epilogue (no autograd)
    if not train:
        with torch.no_grad():
            y2 = o.conv_bn_act(nhwc(x, dtype), None, conv_d, bn_d, spec)
        assert rel_err(nchw(y2), yr.detach()) < tol


@pytest.mark.parametrize("k,h", [(4, 1), (8, 1), (2, 7), (7, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_transpose(k, h, dtype):
    o = ops()
    B, Ci, Co = 3, 32, 16
    ct = torch.nn.ConvTranspose2d(Ci, Co, k, k)
    x = torch.randn(B, Ci, h, h)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
        with torch.no_grad():
            ct.weight.copy_(ct.weight.bfloat16().float())
    xr = x.clone().requires_grad_(True)
    yr = ct(xr)
    probe = torch.randn_like(yr)
    (yr * probe).sum().backward()
    hold = Holder(ct.weight.detach().clone(), ct.bias.detach().clone())
    xd = nhwc(x, torch.float32).requires_grad_(True)     # fp32 input (the to_vec output), cast inside
    y = o.ConvTransposeKS.apply(xd, hold.weight, hold.bias, dtype)
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(nchw(y), yr.detach()) < tol
    (y.float() * nhwc(probe)).sum().backward()
    assert rel_err(nchw(xd.grad), xr.grad) < tol * 2
    assert rel_err(hold.weight.grad.cpu(), ct.weight.grad) < tol * 2
    assert rel_err(hold.bias.grad.cpu(), ct.bias.grad) < tol * 2


@pytest.mark.parametrize("act_name", ["relu", "gelu"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,Cc,H", [(3, 32, 6), (3, 64, 40), (2, 128, 64)], ids=["per-group", "slabs-ragged", "slabs-64x64"])
def test_groupnorm_act(act_name, dtype, B, Cc, H):
    """(3, 32, 6): one workgroup per (sample, group); the larger images take the pixel-slab kernels (ragged last slab at 40x40)."""
    o = ops()
    gn = torch.nn.GroupNorm(8, Cc)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5); gn.bias.uniform_(-0.3, 0.3)
    x = torch.randn(B, Cc, H, H)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    f = F.relu if act_name == "relu" else F.gelu
    yr = f(gn(xr))
    probe = torch.randn_like(yr)
    (yr * probe).sum().backward()
    gw, gb = gn.weight.detach().clone().to(DEV).requires_grad_(True), gn.bias.detach().clone().to(DEV).requires_grad_(True)
    xd = nhwc(x, dtype).requires_grad_(True)
    y = o.GroupNormAct.apply(xd, gw, gb, 8, o.ACT_RELU if act_name == "relu" else o.ACT_GELU)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert rel_err(nchw(y), yr.detach()) < tol
    (y.float() * nhwc(probe)).sum().backward()
    assert rel_err(nchw(xd.grad), xr.grad) < tol * 3
    assert rel_err(gw.grad.cpu(), gn.weight.grad) < tol * 3
    assert rel_err(gb.grad.cpu(), gn.bias.grad) < tol * 3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upcat_film_pool_glue(dtype):
    o = ops()
    B, H = 2, 5
    a, b = torch.randn(B, 16, H, H), torch.randn(B, 8, H, H)
    if dtype == torch.bfloat16:
        a, b = a.bfloat16().float(), b.bfloat16().float()
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.interpolate(torch.cat((ar, br), 1), scale_factor=2, mode="bilinear", align_corners=True)
    probe = torch.randn_like(yr)
    (yr * probe).sum().backward()
    ad, bd = nhwc(a, dtype).requires_grad_(True), nhwc(b, dtype).requires_grad_(True)
    y = o.UpCat.apply(ad, bd)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(nchw(y), yr.detach()) < tol
    (y.float() * nhwc(probe)).sum().backward()
    assert rel_err(nchw(ad.grad), ar.grad) < tol * 2 and rel_err(nchw(bd.grad), br.grad) < tol * 2
    # FiLM
    ce, te = torch.randn(B, 16), torch.randn(B, 16)
    cr, tr = ce.clone().requires_grad_(True), te.clone().requires_grad_(True)
    ar2 = a.clone().requires_grad_(True)
    yr = cr[:, :, None, None] * ar2 + tr[:, :, None, None]
    probe = torch.randn_like(yr)
    (yr * probe).sum().backward()
    ad = nhwc(a, dtype).requires_grad_(True)
    cd, td = ce.to(DEV).requires_grad_(True), te.to(DEV).requires_grad_(True)
    y = o.Film.apply(ad, cd, td)
    assert rel_err(nchw(y), yr.detach()) < tol
    (y.float() * nhwc(probe)).sum().backward()
    assert rel_err(nchw(ad.grad), ar2.grad) < tol * 2
    assert rel_err(cd.grad.cpu(), cr.grad) < tol * 2 and rel_err(td.grad.cpu(), tr.grad) < tol * 2
    # AvgPool k + GELU, MaxPool2, Cat
    x = torch.randn(B, 16, 8, 8)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    xr = x.clone().requires_grad_(True)
    yr = F.gelu(F.avg_pool2d(xr, 4))
    probe = torch.randn_like(yr)
    (yr * probe).sum().backward()
    xd = nhwc(x, dtype).requires_grad_(True)
    y = o.AvgPoolGelu.apply(xd, 4)
    assert rel_err(nchw(y), yr.detach()) < tol
    (y * nhwc(probe)).sum().backward()
    assert rel_err(nchw(xd.grad), xr.grad) < tol * 2
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2)
    probe = torch.randn_like(yr)
    (yr * probe).sum().backward()
    xd = nhwc(x, dtype).requires_grad_(True)
    y = o.MaxPool2.apply(xd)
    assert torch.equal(nchw(y), yr.detach())
    (y.float() * nhwc(probe)).sum().backward()
    assert rel_err(nchw(xd.grad), xr.grad) < tol
    c1, c2 = nhwc(a, dtype).requires_grad_(True), nhwc(b, dtype).requires_grad_(True)
    y = o.Cat.apply(c1, c2)
    assert torch.equal(nchw(y), torch.cat((a, b), 1))
    (y.float() * 2).sum().backward()
    assert torch.all(c1.grad.float() == 2) and torch.all(c2.grad.float() == 2)


def test_linear_act_and_small_gemm_shapes():
    o = ops()
    for (M, K, N) in [(5, 1, 32), (64, 4, 96), (130, 70, 33), (64, 256, 256)]:
        x, w, b = torch.randn(M, K), torch.randn(N, K) / math.sqrt(K), torch.randn(N)
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = F.gelu(F.linear(xr, wr, br))
        probe = torch.randn_like(yr)
        (yr * probe).sum().backward()
        xd, wd, bd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        y = o.Act.apply(o.Linear.apply(xd, wd, bd), o.ACT_GELU)
        assert rel_err(y.cpu(), yr.detach()) < 2e-5
        (y * probe.to(DEV)).sum().backward()
        assert rel_err(xd.grad.cpu(), xr.grad) < 5e-5
        assert rel_err(wd.grad.cpu(), wr.grad) < 5e-5
        assert rel_err(bd.grad.cpu(), br.grad) < 5e-5


@pytest.mark.parametrize("M,K,N", [(2, 4, 8), (64, 1024, 1024), (64, 1024, 64), (64, 64, 1024), (2048, 128, 8), (2048, 8, 128),
                                   (200, 20, 36), (1024, 16, 16), (77, 512, 4), (512, 32, 512)])
def test_dense_layer_kernels_exact_integers(M, K, N):
    """dense.hip behind dm_linear_fwd / dm_linear_bwd (SE MLP, CoordAttn strip convolutions, EmbedFC): small-integer fp32 data,
    every product and partial sum exact, so y, dx, dw (accumulated onto a non-zero start, rows split over workgroups with
    atomics) and db must equal the CPU result bit for bit — ragged M / K / N (K, N multiples of 4) included."""
    from diffusionmodel_amd._lib import call, ptr
    g = torch.Generator().manual_seed(M + 3 * K + 7 * N)
    ri = lambda *s: torch.randint(-2, 3, s, generator=g).float()
    keep = lambda t, p: t * (torch.rand(t.shape, generator=g) < p).float()       # sparse: |sums| stay far below 2^24
    x, w, b, gy = keep(ri(M, K), 0.3), keep(ri(N, K), 0.3), ri(N), keep(ri(M, N), 0.3)
    dw0, db0 = ri(N, K), ri(N)
    xd, wd, bd, gd = x.to(DEV), w.to(DEV), b.to(DEV), gy.to(DEV)
    y = torch.full((M, N), 7.0, device=DEV)
    call("dm_linear_fwd", ptr(xd), ptr(wd), ptr(bd), ptr(y), M, K, N, 0)
    assert torch.equal(y.cpu(), x @ w.t() + b)
    y2 = torch.empty((M, N), device=DEV)
    call("dm_linear_fwd", ptr(xd), ptr(wd), None, ptr(y2), M, K, N, 1)            # no bias, fused GELU (DM_ACT_GELU)
    assert torch.allclose(y2.cpu(), F.gelu(x @ w.t()), rtol=1e-6, atol=1e-6)
    dx, dw, db = torch.full((M, K), -3.0, device=DEV), dw0.to(DEV), db0.to(DEV)
    call("dm_linear_bwd", ptr(xd), ptr(wd), ptr(gd), ptr(dx), ptr(dw), ptr(db), M, K, N)
    assert torch.equal(dx.cpu(), gy @ w)
    assert torch.equal(dw.cpu(), dw0 + gy.t() @ x)
    assert torch.equal(db.cpu(), db0 + gy.sum(0))
    dx2 = torch.empty((M, K), device=DEV)
    call("dm_linear_bwd", ptr(xd), ptr(wd), ptr(gd), ptr(dx2), None, None, M, K, N)   # input gradient only
    assert torch.equal(dx2, dx)


@pytest.mark.parametrize("N,coff,ldc", [(128, 128, 256), (64, 8, 136), (128, 4, 132)])
def test_conv_output_into_channel_slice(N, coff, ldc):
    """DmConv.ldc / coff (write the N output channels at column `coff` of rows `ldc` wide): the 16-byte-store epilogue (coff and
    ldc multiples of 8) and the 8-byte one (coff = 4) must both leave the other columns alone and equal the dense output."""
    o = ops()
    from diffusionmodel_amd import ops as O
    B, S, Ci = 2, 16, 64
    g = torch.Generator().manual_seed(N + coff)
    x = torch.randn(B, S, S, Ci, generator=g).to(DEV).bfloat16()
    conv = Holder((torch.randn(N, Ci, 3, 3, generator=g) * 0.05), torch.randn(N, generator=g))
    spec = o.ConvSpec(3, 3, 1, 1)
    with torch.no_grad():
        ref = o.conv_bn_act(x, None, conv, None, spec)                       # (B, S, S, N)
        wp = O.packed_fwd(conv.weight, torch.bfloat16, Ci)
        wide = torch.full((B, S, S, ldc), 7.0, device=DEV, dtype=torch.bfloat16)
        O._conv_call(x, None, O.ptr(wp), 9 * Ci, wide, dtype=torch.bfloat16, B=B, Hi=S, Wi=S, C1=Ci, C2=0, Hq=S, Wq=S, sy=1, sx=1, T=9, KW=3,
                     ty=1, tx=1, oy0=-1, ox0=-1, Ho=S, Wo=S, N=N, ldc=ldc, coff=coff, shift=conv.bias)
    assert torch.equal(wide[..., coff:coff + N], ref)
    assert torch.all(wide[..., :coff] == 7.0) and torch.all(wide[..., coff + N:] == 7.0)


def test_broadcast_second_source_equals_repeated_batch():
    """CFG sampler: skip tensors of n samples under a batch of 2n (DmConv.in2_batch, dm_upcat_fwd_bcast) — bit-identical to
    feeding the repeated tensor, and refused where it cannot work (autograd, a gather-kernel shape)."""
    o = ops()
    n, F, S = 4, 64, 32
    g = torch.Generator().manual_seed(9)
    x1 = torch.randn(2 * n, S, S, F, generator=g).to(DEV).bfloat16()
    x2 = torch.randn(n, S, S, F, generator=g).to(DEV).bfloat16()
    with torch.no_grad():
        a = o.UpCat.apply(x1, x2)
        b = o.UpCat.apply(x1, torch.cat([x2, x2]))
        assert torch.equal(a, b)
        conv = Holder((torch.randn(F, 2 * F, 3, 3, generator=g) * 0.05).to(DEV), torch.randn(F, generator=g).to(DEV))
        spec = o.ConvSpec(3, 3, 1, 1)
        ya = o.conv_bn_act(x1, x2, conv, None, spec)
        yb = o.conv_bn_act(x1, torch.cat([x2, x2]), conv, None, spec)
        assert torch.equal(ya, yb)
        assert o.conv_bcast_ok(torch.bfloat16, 2 * n, S, S, F, F) and not o.conv_bcast_ok(torch.float32, 2 * n, S, S, F, F)
        with pytest.raises(Exception):                       # 32 channels: not a halo-kernel shape -> the library refuses
            o.conv_bn_act(x1[..., :32].contiguous(), x2[..., :32].contiguous(),
                          Holder(torch.randn(F, 64, 3, 3).to(DEV), torch.randn(F).to(DEV)), None, spec)
    with pytest.raises(Exception):                           # training graph: inference-only
        o.UpCat.apply(x1.clone().requires_grad_(True), x2)


def test_loss_qsample_cfg_update_randn():
    o = ops()
    from oracle import unet_ref as O
    B, Cc, S = 3, 3, 16
    pred, noise = torch.randn(B, Cc, S, S), torch.randn(B, Cc, S, S)
    mask = torch.full((B, S, S), 0.5)
    mask[:, 8:] = 1.0
    mask[:, 2:6, 3:9] = 3.0
    pr = pred.clone().requires_grad_(True)
    lr = O.weighted_loss(pr, noise, mask)
    (lr * 0.25).backward()
    cfg6 = torch.tensor([1.2, 0.8, 3.0, 1.0, 0.5, 2.0], device=DEV)
    pd = pred.to(DEV).requires_grad_(True)
    ld = o.WeightedLoss.apply(pd, noise.to(DEV), mask.to(DEV), cfg6)
    assert abs(ld.item() - lr.item()) < 2e-6 * max(1, abs(lr.item()))
    (ld * 0.25).backward()
    assert rel_err(pd.grad.cpu(), pr.grad) < 1e-5
    # plain MSE (MNIST)
    pd2 = pred.to(DEV).requires_grad_(True)
    l2 = o.WeightedLoss.apply(pd2, noise.to(DEV), None, None)
    assert abs(l2.item() - F.mse_loss(pred, noise).item()) < 1e-5
    l2.backward()
    assert rel_err(pd2.grad.cpu(), 2 * (pred - noise) / pred.numel()) < 1e-5
    # q-sample
    sched = O.ddpm_schedules(1e-4, 0.02, 1000)
    ts = torch.tensor([1, 500, 1000])
    xt_r = O.q_sample(sched, pred, ts, noise)
    xt = o.qsample(pred.to(DEV), noise.to(DEV), ts.to(DEV), sched["sqrtab"].to(DEV), sched["sqrtmab"].to(DEV), torch.float32, 8)
    assert xt.shape == (B, S, S, 8)
    assert torch.equal(xt[..., 3:].cpu(), torch.zeros(B, S, S, 5))
    assert rel_err(nchw(xt[..., :3]), xt_r) < 1e-6
    # cfg update: injected z, i > 1 and i == 1 (z ignored), then step counter decremented
    n = 2
    x = torch.randn(n, Cc, S, S)
    eps = torch.randn(2 * n, Cc, S, S)
    z = torch.randn(n, Cc, S, S)
    sd = {k: v.to(DEV) for k, v in sched.items()}
    for i in (1000, 2, 1):
        step = torch.tensor([i], dtype=torch.int32, device=DEV)
        xd = x.to(DEV).clone()
        o.cfg_update(xd, eps.to(DEV), z.to(DEV), 2.0, sd, step)
        xr = O.cfg_update(sched, i, x, eps[:n], eps[n:], 2.0, z if i > 1 else torch.zeros_like(z))
        assert rel_err(xd.cpu(), xr) < 2e-6, i
        assert int(step.item()) == i - 1
    # Philox N(0,1): moments, determinism, offset independence
    r1 = o.randn((1 << 20,), DEV, 1234, 7)
    r2 = o.randn((1 << 20,), DEV, 1234, 7)
    r3 = o.randn((1 << 20,), DEV, 1234, 8)
    assert torch.equal(r1, r2) and not torch.equal(r1, r3)
    assert abs(r1.mean().item()) < 5e-3 and abs(r1.std().item() - 1) < 5e-3
    assert abs((r1 * r3).mean().item()) < 5e-3
    assert abs((r1 ** 4).mean().item() - 3.0) < 0.05


def test_fused_adamw_matches_torch():
    from diffusionmodel_amd.optim import FusedAdamW
    torch.manual_seed(1)
    shapes = [(8, 4, 3, 3), (8,), (5, 7), (1,), (16, 8, 1, 1)]
    ref = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    dev = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref]
    opt_r = torch.optim.AdamW(ref, lr=1e-2, weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-8)
    opt_d = FusedAdamW(dev, lr=1e-2, weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0)
    for it in range(3):
        grads = [torch.randn(s) * (3.0 if it == 0 else 0.05) for s in shapes]   # first step clips, later ones do not
        for p, g in zip(ref, grads):
            p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        opt_r.step()
        opt_d.zero_grad()
        for p, g in zip(dev, grads):
            p.grad = g.to(DEV)
        opt_d.step()
        for p, q in zip(ref, dev):
            assert rel_err(q.detach().cpu(), p.detach()) < 2e-6, it


@pytest.mark.parametrize("M,Cc", [(32, 4), (32, 8), (8, 16), (24, 2), (600, 12), (1100, 64)])
@pytest.mark.parametrize("train", [True, False])
def test_bn_act_matrix_strips(M, Cc, train):
    """BatchNorm(+GELU) over the rows of small fp32 matrices (CoordAttn strips): vector and scalar column paths."""
    o = ops()
    bn_r = torch.nn.BatchNorm1d(Cc)
    with torch.no_grad():
        bn_r.weight.uniform_(0.5, 1.5); bn_r.bias.uniform_(-0.3, 0.3)
        bn_r.running_mean.uniform_(-0.2, 0.2); bn_r.running_var.uniform_(0.5, 1.5)
    bn_d = torch.nn.BatchNorm2d(Cc).to(DEV)
    bn_d.load_state_dict(bn_r.state_dict())
    bn_r.train(train); bn_d.train(train)
    z = torch.randn(M, Cc) * 0.7 + 0.3
    zr = z.clone().requires_grad_(True)
    yr = F.gelu(bn_r(zr))
    probe = torch.randn_like(yr)
    (yr * probe).sum().backward()
    zd = z.to(DEV).requires_grad_(True)
    spec = o.ConvSpec(1, 1, 1, 0, o.ACT_GELU, bn_d)
    y = o.BnActMatrix.apply(zd, bn_d.weight, bn_d.bias, bn_d, spec, o.ACT_GELU)
    assert rel_err(y.cpu(), yr.detach()) < 1e-5
    (y * probe.to(DEV)).sum().backward()
    assert rel_err(zd.grad.cpu(), zr.grad) < 5e-5
    assert rel_err(bn_d.weight.grad.cpu(), bn_r.weight.grad) < 5e-5
    assert rel_err(bn_d.bias.grad.cpu(), bn_r.bias.grad) < 5e-5
    if train:
        assert rel_err(bn_d.running_var.cpu(), bn_r.running_var) < 1e-5


def test_batched_pack_device_rng_offset_and_paired_reduce():
    """The launch-saving entry points against the single-purpose ones they replace:
    dm_pack_multi == dm_pack_wT per tensor, dm_randn_dev(offset on device) == dm_randn(offset), dm_col_reduce2 == 2 x dm_col_reduce."""
    import ctypes as C
    from diffusionmodel_amd._lib import call, ptr, DM_BF16, DM_F32
    # ---- dm_pack_multi: two tensors, different tap subsets / dtypes
    # N, T, C, taps, Np, dtype, source = bf16 shadow (flag 0x100)?   (ragged against the 64 x 64 tiles)
    specs = [(70, 9, 100, [0, 2, 4, 8], 72, torch.bfloat16, False), (7, 4, 16, [1, 3], 8, torch.float32, False),
             (130, 9, 64, list(range(9)), 136, torch.bfloat16, True)]
    srcs, refs, outs, ents, taps_rows, blocks = [], [], [], [], [], []
    for e, (N, T, Cc, taps, Np, dtp, shadow) in enumerate(specs):
        src = torch.randn(N, T, Cc, device=DEV)
        if shadow:
            src = src.bfloat16().float()
        ref = torch.empty(Cc, len(taps), Np, device=DEV, dtype=dtp)
        out = torch.full_like(ref, 7.0)
        code = DM_BF16 if dtp == torch.bfloat16 else DM_F32
        call("dm_pack_wT", ptr(src), ptr(ref), code, N, T, Cc, len(taps), (C.c_int32 * len(taps))(*taps), Np)
        src_in = src.bfloat16() if shadow else src
        ents.append((src_in.data_ptr(), out.data_ptr(), N, T, Cc, len(taps), Np, code | (0x100 if shadow else 0)))
        taps_rows.append(taps + [0] * (16 - len(taps)))
        blocks += [(e, nt, ct, tt) for tt in range(len(taps)) for ct in range((Cc + 63) // 64) for nt in range((Np + 63) // 64)]
        srcs.append(src_in); refs.append(ref); outs.append(out)
    t_e = torch.tensor(ents, dtype=torch.int64).to(DEV)
    t_t = torch.tensor(taps_rows, dtype=torch.int32).to(DEV)
    t_b = torch.tensor(blocks, dtype=torch.int32).to(DEV)
    call("dm_pack_multi", ptr(t_e), ptr(t_t), ptr(t_b), len(blocks))
    for ref, out in zip(refs, outs):
        assert torch.equal(ref, out)
    # ---- dm_randn_dev
    a, b = torch.empty(1000, device=DEV), torch.empty(1000, device=DEV)
    off = torch.tensor([12345], dtype=torch.int64, device=DEV)
    call("dm_randn", ptr(a), 1000, 77, 12345)
    call("dm_randn_dev", ptr(b), 1000, 77, ptr(off))
    assert torch.equal(a, b)
    # ---- dm_col_reduce2
    p1, p2 = torch.randn(37, 20, device=DEV), torch.randn(37, 20, device=DEV)
    o1, o2, r1, r2 = (torch.empty(20, device=DEV) for _ in range(4))
    call("dm_col_reduce", ptr(p1), 37, 20, ptr(r1), 0)
    call("dm_col_reduce", ptr(p2), 37, 20, ptr(r2), 0)
    call("dm_col_reduce2", ptr(p1), ptr(p2), 37, 20, ptr(o1), ptr(o2))
    assert torch.equal(o1, r1) and torch.equal(o2, r2)


TAP4_CASES = [
    # B, C, Cout, H (= W, input): output rows of H/2 pixels
    (2, 64, 128, 64),        # rows of 32: four tiles per image, one chunk per parity sub-image
    (1, 128, 192, 32),       # rows of 16, two chunks per sub-image, ragged N (two n-tiles)
    (4, 64, 64, 16),         # 8x8 output: four images per tile, N = 64
    (8, 256, 128, 16),       # 8x8 output, 16 chunks over 2 tiles: the channel chunks are split over workgroups (split-K epilogue)
    (1, 64, 128, 128),       # rows of 64
    (2, 128, 136, 64),       # ragged N in the second n-tile
    (16, 64, 128, 64),       # enough tiles that the four parity classes of the input gradient go out as ONE launch (dm_conv_parity4)
    (256, 64, 64, 16),       # the same with 8x8 outputs (four images per tile)
]


@pytest.mark.parametrize("case", TAP4_CASES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_conv4x4s2_four_tap_halo_kernel_exact_integers(case, dtype):
    """The 4x4 / stride-2 / pad-1 convolution of UnetDown (new_scripy.py:229) on the four-tap halo kernel (conv_tap4_halo_kernel:
    forward over the parity sub-images, input gradient one output-parity class per launch with a strided output) — small-integer
    data, so forward, input gradient and weight gradient must equal F.conv2d bit for bit."""
    o = ops()
    from diffusionmodel_amd import _lib
    lib = _lib.load()
    B, C, Co, H = case
    g = torch.Generator().manual_seed(C + Co + H)
    ri = lambda *s: torch.randint(-1, 2, s, generator=g).float()
    x, w, b, probe = ri(B, C, H, H), ri(Co, C, 4, 4), ri(Co), ri(B, Co, H // 2, H // 2)
    w = w * (torch.rand(Co, C, 4, 4, generator=g) < 0.15).float()            # sparse weights keep |y| < 256 (exact in 16 bits)
    probe = probe * (torch.rand(B, Co, H // 2, H // 2, generator=g) < 0.3).float()
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=2, padding=1)
    assert yr.abs().max() < 256
    (yr * probe).sum().backward()
    assert xr.grad.abs().max() < 256
    conv = Holder(w, b)
    xd = nhwc(x, dtype).requires_grad_(True)
    y = o.conv_bn_act(xd, None, conv, None, o.ConvSpec(4, 4, 2, 1))
    assert lib.dm_last_conv_path() == 2, "the forward launch did not take the four-tap halo kernel"
    assert torch.equal(nchw(y), yr.detach())
    (y.float() * nhwc(probe)).sum().backward()
    # (the input gradient reduces over Cout: whole 64-channel chunks go to the four-tap kernel, a ragged Cout to the gather kernel)
    assert lib.dm_last_conv_path() == (2 if Co % 64 == 0 else 0), "unexpected kernel for the input-gradient launches"
    assert torch.equal(nchw(xd.grad), xr.grad)
    # the weight gradient: four-tap (S2) form of the halo-resident kernel on output rows of 32 / 16 pixels and 8x8 output images
    assert lib.dm_last_wgrad_path() == (3 if H // 2 in (8, 16, 32) else 0), "unexpected kernel for the weight gradient"
    assert torch.equal(conv.weight.grad.cpu(), wr.grad)
    assert torch.equal(conv.bias.grad.cpu(), br.grad)


def test_splitk_last_arriver_form_is_bit_exact_too():
    """dm_set_splitk_inkernel(1) — the default since r03: the split that arrives last folds the others' partials and runs the epilogue in
    the same launch (sc1 stores / agent-scope counter / sc1 loads, no fence: igemm_dev.h); 0 = the two-launch form, also exact."""
    from diffusionmodel_amd import _lib
    lib = _lib.load()
    o = ops()
    assert lib.dm_set_splitk_inkernel(1) == 0
    try:
        for (B, C, Co, H) in ((4, 256, 128, 8), (2, 512, 256, 16)):
            g = torch.Generator().manual_seed(B + C)
            ri = lambda *s: torch.randint(-1, 2, s, generator=g).float()
            x, w, b = ri(B, C, H, H), ri(Co, C, 3, 3) * (torch.rand(Co, C, 3, 3, generator=g) < 0.1).float(), ri(Co)
            yr = F.conv2d(x, w, b, padding=1)
            assert yr.abs().max() < 256
            conv = Holder(w, b)
            for rep in range(3):                                  # the arrival counters must be back at zero after every launch
                y = o.conv_bn_act(nhwc(x, torch.bfloat16), None, conv, None, o.ConvSpec(3, 3, 1, 1))
                assert torch.equal(nchw(y), yr)
            assert lib.dm_set_splitk_inkernel(0) == 0             # the two-launch form on the same problem
            y = o.conv_bn_act(nhwc(x, torch.bfloat16), None, conv, None, o.ConvSpec(3, 3, 1, 1))
            assert torch.equal(nchw(y), yr)
            assert lib.dm_set_splitk_inkernel(1) == 0
    finally:
        assert lib.dm_set_splitk_inkernel(1) == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("k,B,C,N,H", [(3, 2, 64, 128, 32), (1, 2, 32, 128, 16), (3, 1, 64, 72, 16), (1, 3, 64, 40, 8)])
def test_conv_epilogue_addend_exact_integers(dtype, k, B, C, N, H):
    """DmConv.addend (the gradient of a tensor with two consumers, ops.GradFork): out = conv(x) + addend, exact on small integers —
    the halo kernel and the gather kernel, the 16-byte store path (addend fetched at the store address and swapped back to the
    accumulator layout) and the narrow paths (ragged N, fp32)."""
    o = ops()
    g = torch.Generator().manual_seed(k * 100 + C + N)
    ri = lambda *s: torch.randint(-2, 3, s, generator=g).float()
    x, w, add = ri(B, C, H, H), ri(N, C, k, k) * (torch.rand(N, C, k, k, generator=g) < 0.2).float(), ri(B, N, H, H) * 3
    ref = F.conv2d(x, w, None, padding=k // 2) + add
    assert ref.abs().max() < 256
    xd, ad = nhwc(x, dtype), nhwc(add, dtype)
    wp = o.packed_fwd(Holder(w, None).weight, dtype, C)
    out = torch.empty(B, H, H, N, dtype=dtype, device=DEV)
    o._conv_call(xd, None, wp.data_ptr(), k * k * C, out, dtype=dtype, B=B, Hi=H, Wi=H, C1=C, C2=0, Hq=H, Wq=H, sy=1, sx=1, T=k * k, KW=k, ty=1, tx=1,
                 oy0=-(k // 2), ox0=-(k // 2), Ho=H, Wo=H, N=N, addend=ad)
    assert torch.equal(nchw(out), ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,C,N,H", [(2, 32, 128, 16), (1, 64, 64, 12), (2, 128, 32, 8), (3, 128, 256, 10), (1, 32, 40, 9), (2, 64, 136, 16)])
def test_pointwise_conv_kernel_exact_integers(dtype, B, C, N, H):
    """conv_pw_kernel (1x1, K in {32, 64, 128}: fragments straight from global memory) forward and — through the transposed pack —
    as the input gradient of a 1x1 layer whose output width is 32 / 64 / 128; ragged M (72 .. 300 rows) and N; bit-exact on integers."""
    o = ops()
    from diffusionmodel_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 1000 + C + N)
    ri = lambda *s: torch.randint(-2, 3, s, generator=g).float()
    x, w, b, probe = ri(B, C, H, H), ri(N, C, 1, 1), ri(N), ri(B, N, H, H)
    w = w * (torch.rand(N, C, 1, 1, generator=g) < 0.5).float()
    probe = probe * (torch.rand(B, N, H, H, generator=g) < 0.3).float()
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br)
    assert yr.abs().max() < 256
    (yr * probe).sum().backward()
    assert xr.grad.abs().max() < 256
    conv = Holder(w, b)
    xd = nhwc(x, dtype).requires_grad_(True)
    y = o.conv_bn_act(xd, None, conv, None, o.ConvSpec(1, 1, 1, 0))
    assert lib.dm_last_conv_path() == 3, "the forward launch did not take the pointwise kernel"
    assert torch.equal(nchw(y), yr.detach())
    (y.float() * nhwc(probe)).sum().backward()
    assert torch.equal(nchw(xd.grad), xr.grad)
    assert torch.equal(conv.weight.grad.cpu(), wr.grad)
    assert torch.equal(conv.bias.grad.cpu(), br.grad)


def test_pointwise_conv_kernel_batchnorm_statistics():
    """Train-mode BatchNorm behind the pointwise kernel (UnetDown.channel_compress at n_feat = 128: 128 -> 32 channels + BN + GELU)."""
    o = ops()
    torch.manual_seed(0)
    B, Ci, Co, H = 4, 128, 32, 16
    x = torch.randn(B, Ci, H, H).bfloat16().float()
    conv_r = torch.nn.Conv2d(Ci, Co, 1)
    with torch.no_grad():
        conv_r.weight.copy_(conv_r.weight.bfloat16().float())
    bn_r, bn_d = torch.nn.BatchNorm2d(Co), torch.nn.BatchNorm2d(Co).to(DEV)
    with torch.no_grad():
        bn_r.weight.uniform_(0.5, 1.5); bn_r.bias.uniform_(-0.3, 0.3)
    bn_d.load_state_dict(bn_r.state_dict())
    conv_d = Holder(conv_r.weight.detach().clone(), conv_r.bias.detach().clone())
    xr = x.clone().requires_grad_(True)
    yr = F.gelu(bn_r(conv_r(xr)))
    probe = torch.randn_like(yr)
    (yr * probe).sum().backward()
    xd = nhwc(x, torch.bfloat16).requires_grad_(True)
    y = o.conv_bn_act(xd, None, conv_d, bn_d, o.ConvSpec(1, 1, 1, 0, o.ACT_GELU, bn_d))
    assert rel_err(nchw(y), yr.detach()) < 1e-2
    assert rel_err(bn_d.running_mean.cpu(), bn_r.running_mean) < 1e-2 and rel_err(bn_d.running_var.cpu(), bn_r.running_var) < 1e-2
    (y.float() * nhwc(probe)).sum().backward()
    assert rel_err(nchw(xd.grad), xr.grad) < 1.5e-2
    assert rel_err(conv_d.weight.grad.cpu(), conv_r.weight.grad) < 1.5e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,C,N,H", [
    (2, 32, 128, 16), (3, 128, 32, 10),          # 128 x 32 block, 128-pixel chunks; the second with the input as the wide operand and a ragged last chunk
    (4, 32, 256, 32), (1, 256, 32, 9),           # 256 x 32
    (2, 64, 512, 16), (3, 256, 64, 8),           # 256 x 64 blocks (two along N / one along C)
    (2, 128, 128, 16), (1, 128, 384, 12), (3, 512, 128, 8),   # 128 x 128 blocks: square, three along N, four along C
])
def test_pointwise_wgrad_kernel_exact_integers(dtype, B, C, N, H):
    """wgrad_pw_kernel (weight gradient of the 1x1 layers: one of N, C in {32, 64, 128}, the other in whole blocks): every block
    shape with either operand as the wide one, several pixel splits meeting through the atomics, ragged last chunks (81 .. 300
    rows), bias gradient from either image; bit-exact on integers."""
    o = ops()
    from diffusionmodel_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 1000 + C + N)
    ri = lambda *s: torch.randint(-2, 3, s, generator=g).float()
    x, w, b, probe = ri(B, C, H, H), ri(N, C, 1, 1), ri(N), ri(B, N, H, H)
    w = w * (torch.rand(N, C, 1, 1, generator=g) < 0.1).float()
    probe = probe * (torch.rand(B, N, H, H, generator=g) < 0.3).float()
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br)
    assert yr.abs().max() < 256
    (yr * probe).sum().backward()
    assert xr.grad.abs().max() < 256
    conv = Holder(w, b)
    xd = nhwc(x, dtype).requires_grad_(True)
    y = o.conv_bn_act(xd, None, conv, None, o.ConvSpec(1, 1, 1, 0))
    assert torch.equal(nchw(y), yr.detach())
    try:
        (y.float() * nhwc(probe)).sum().backward()
        assert lib.dm_last_wgrad_path() == 2, "the weight gradient did not take the 1x1 kernel"
        assert torch.equal(conv.weight.grad.cpu(), wr.grad)
        assert torch.equal(conv.bias.grad.cpu(), br.grad)
        assert torch.equal(nchw(xd.grad), xr.grad)
        for knob in ((1, 96, 0), (0, 0, 0)):          # same result from a different split count (96 workgroups) and from the per-tap kernel
            assert lib.dm_set_wgrad_pw(*knob) == 0
            conv.weight.grad = None
            conv.bias.grad = None
            y = o.conv_bn_act(nhwc(x, dtype).requires_grad_(True), None, conv, None, o.ConvSpec(1, 1, 1, 0))
            (y.float() * nhwc(probe)).sum().backward()
            assert lib.dm_last_wgrad_path() == (2 if knob[0] else 0)
            assert torch.equal(conv.weight.grad.cpu(), wr.grad)
    finally:
        assert lib.dm_set_wgrad_pw(1, 384, 0) == 0


# ---- fused attention chains (chain.hip): against the one-launch-per-op path AND torch on the CPU --------------------------------
@pytest.mark.parametrize("B,C,HW,dtype", [(64, 128, 4096, torch.bfloat16), (5, 1024, 64, torch.bfloat16), (20, 1536, 16, torch.float32),
                                          (2, 32, 256, torch.float32), (33, 256, 1024, torch.float16), (16, 2048, 64, torch.float32)])
def test_fused_se_chain_matches_unfused_and_torch(B, C, HW, dtype):
    """SeResidual = (res + x2 * sigmoid(W2 gelu(W1 mean_hw(x2)))) / 1.414 (new_scripy.py:143-158, 196-205): the fused chain
    (dm_se_fwd / dm_se_bwd: pooling partials folded inside the MLP kernel, hidden vector in LDS) against the r02 path
    (pool, fold, 2 dense, 2 activation launches) and against torch fp32 on the CPU — forward, dx2, dres, dW1, dW2."""
    from diffusionmodel_amd import ops as o
    torch.manual_seed(C + B)
    R = C // 16
    H = int(math.isqrt(HW))
    W = HW // H
    x2 = torch.randn(B, H, W, C).to(dtype)
    res = torch.randn(B, H, W, C).to(dtype)
    w1 = torch.randn(R, C) / math.sqrt(C)
    w2 = torch.randn(C, R) / math.sqrt(R)
    probe = torch.randn(B, H, W, C)
    inv = 1.0 / 1.414

    def run(fused):
        o.FUSED_CHAINS = fused
        a, b = x2.to(DEV).requires_grad_(True), res.to(DEV).requires_grad_(True)
        p1, p2 = torch.nn.Parameter(w1.to(DEV)), torch.nn.Parameter(w2.to(DEV))
        out = o.SeResidual.apply(a, b, p1, p2, inv, True)
        (out.float() * probe.to(DEV)).sum().backward()
        with torch.no_grad():
            out_ng = o.SeResidual.apply(x2.to(DEV), res.to(DEV), p1, p2, inv, False)
        return [t.detach().float().cpu() for t in (out, a.grad, b.grad, p1.grad, p2.grad, out_ng)]
    try:
        fu, un = run(True), run(False)
    finally:
        o.FUSED_CHAINS = True
    # torch reference on the values the kernels see (the 16-bit inputs, fp32 math)
    a, b = x2.float().requires_grad_(True), res.float().requires_grad_(True)
    p1, p2 = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    y = a.mean(dim=(1, 2))
    sg = torch.sigmoid(F.gelu(y @ p1.t()) @ p2.t())
    out = (b + a * sg[:, None, None, :]) * inv
    (out * probe).sum().backward()
    ref = [out.detach(), a.grad, b.grad, p1.grad, p2.grad, out.detach()]
    tol = 2e-5 if dtype == torch.float32 else 1.2e-2
    for name, f_, u_, r_ in zip(("out", "dx2", "dres", "dW1", "dW2", "out(no grad)"), fu, un, ref):
        scale = float(r_.abs().max()) + 1e-12
        ef, eu = float((f_ - r_).abs().max()) / scale, float((u_ - r_).abs().max()) / scale
        assert ef <= max(tol, 1.5 * eu), (name, ef, eu)
        assert float((f_ - u_).abs().max()) / scale <= tol, (name, float((f_ - u_).abs().max()) / scale)


@pytest.mark.parametrize("B,C,H,W", [(4, 64, 8, 8), (8, 128, 32, 32), (16, 1024, 4, 4), (3, 128, 16, 24), (2, 2048, 8, 8), (5, 192, 6, 10),
                                     (2, 64, 40, 17)])
@pytest.mark.parametrize("train", [False, True])
def test_fused_coordattn_chain_matches_the_oracle_and_the_unfused_path(B, C, H, W, train):
    """CoordAttn (new_scripy.py:70-140) with the strip chain fused (dm_ca_chain_fwd / _bwd: conv1 + BatchNorm + GELU, the cross
    projections, the gated mix with adaptive pooling between strips of different length, conv_h / conv_w) in fp32: output, input
    gradient and every parameter gradient against the float64 oracle (which follows the reference line by line, adaptive pools
    included: H != W is covered — VERDICT r02 missing #4) and, where the one-launch-per-op path exists (H == W), against it."""
    import diffusionmodel_amd as D
    from diffusionmodel_amd import ops as o
    from oracle import unet_ref as O
    torch.manual_seed(1000 + C + H)
    mod = D.CoordAttn(C).to(DEV)
    with torch.no_grad():
        for n_, p_ in mod.named_parameters():
            if p_.dim() == 1 and p_.numel() == 1:
                p_.fill_(0.3 if "gamma" in n_ else -0.2)
            elif "bn1" in n_:
                p_.uniform_(0.5, 1.5) if n_.endswith("weight") else p_.uniform_(-0.2, 0.2)
        for n_, b_ in mod.named_buffers():
            if n_.endswith("running_mean"):
                b_.normal_(0, 0.1)
            elif n_.endswith("running_var"):
                b_.uniform_(0.6, 1.4)
    sd = {k: v.detach().clone().cpu() for k, v in mod.state_dict().items()}
    x = torch.randn(B, C, H, W)
    probe = torch.randn(B, C, H, W)

    def run(fused):
        o.FUSED_CHAINS = fused
        mod.load_state_dict(sd)
        mod.train(train)
        mod.zero_grad()
        xd = x.to(DEV).requires_grad_(True)
        y = mod(xd)
        (y * probe.to(DEV)).sum().backward()
        out = {"y": y.detach().cpu(), "dx": xd.grad.cpu()}
        out.update({"d." + n_: p_.grad.detach().cpu().reshape(-1) for n_, p_ in mod.named_parameters()})
        out.update({"buf." + n_: b_.detach().cpu().float() for n_, b_ in mod.named_buffers() if "running" in n_})
        with torch.no_grad():
            mod.load_state_dict(sd)
            mod.train(train)
            out["y_nograd"] = mod(x.to(DEV)).cpu()
        return out
    try:
        fu = run(True)
        un = run(False) if H == W else None
    finally:
        o.FUSED_CHAINS = True
    P = {"ca." + k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone().double() if v.is_floating_point() else v.clone())
         for k, v in sd.items()}
    xr = x.double().requires_grad_(True)
    yr = O.coord_attn(xr, P, "ca", train)
    (yr * probe.double()).sum().backward()
    ref = {"y": yr.detach(), "dx": xr.grad, "y_nograd": yr.detach()}
    ref.update({"d." + k[3:]: v.grad.reshape(-1) for k, v in P.items() if v.is_floating_point() and v.requires_grad})
    if train:
        ref.update({"buf." + k[3:]: v for k, v in P.items() if "running" in k})
    worst = {}
    for k, r_ in ref.items():
        scale = float(r_.abs().max()) + 1e-12
        if train and k in ("d.conv1_h.bias", "d.conv1_w.bias"):
            # train-mode BatchNorm removes the batch mean: this gradient is zero in exact arithmetic (the float64 value is ~1e-17), so
            # it is held to the scale of the same layer's weight gradient instead of to itself
            scale = float(ref[k.replace("bias", "weight")].abs().max()) + 1e-12
        ef = float((fu[k].double() - r_).abs().max()) / scale
        worst[k] = ef
        bar = 2e-4 if not train else 2e-3          # fp32 kernels vs float64; train-mode BatchNorm over few rows amplifies rounding
        if k in ("d.alpha", "d.beta", "d.gamma_h", "d.gamma_w"):
            # scalar gradients: ONE sum over B*C*H*W products of mixed sign, measured against the (cancelled) total — the fused
            # chain folds it through fp32 atomics whose order changes from run to run (seen: 1.2e-4 and 2.1e-4 on the same case)
            bar = max(bar, 1e-3)
        if un is not None:
            eu = float((un[k].double() - r_).abs().max()) / scale
            assert ef <= max(bar, 3 * eu), (k, ef, eu)
        else:
            assert ef <= bar, (k, ef)
    print(f"CoordAttn B{B} C{C} {H}x{W} train={train}: worst rel err vs float64 oracle {max(worst.values()):.2e} ({max(worst, key=worst.get)})")


# ---- persistent halo kernel (igemm_halo_p.hip): cross-tile prefetch, counted waits over the epilogue's stores ----------------------
@pytest.mark.parametrize("B,H,C,N,dtype", [(52, 64, 64, 128, torch.bfloat16), (50, 64, 128, 128, torch.float16), (100, 32, 128, 256, torch.bfloat16),
                                           (264, 16, 192, 384, torch.bfloat16), (620, 8, 64, 640, torch.bfloat16), (13, 128, 64, 128, torch.bfloat16)])
def test_persistent_halo_kernel_is_bit_identical_to_one_workgroup_per_tile(B, H, C, N, dtype):
    """conv3x3_halo_pkernel walks over several tiles per workgroup and fetches the next tile's halo / weights behind the current
    tile's last chunk; the k order of a tile is the one-workgroup-per-tile kernel's, so forward, input gradient (with and without
    the forked-gradient addend), BatchNorm statistics and weight gradient must agree BIT FOR BIT on random data; one integer case
    is also held to F.conv2d."""
    o = ops()
    from diffusionmodel_amd import _lib as L
    lib = L.load()
    torch.manual_seed(B + H + C)
    x = torch.randn(B, H, H, C).to(dtype).to(DEV)
    w = torch.nn.Parameter((torch.randn(N, C, 3, 3) / math.sqrt(9 * C)).to(DEV).contiguous(memory_format=torch.channels_last))
    bias = torch.nn.Parameter(torch.randn(N).to(DEV) * 0.1)
    probe = torch.randn(B, H, H, N).to(dtype).to(DEV)
    bn = torch.nn.BatchNorm2d(N).to(DEV)
    bn.train()
    sd = {k: v.clone() for k, v in bn.state_dict().items()}

    def run(persist, with_bn):
        lib.dm_set_conv_persist(persist)
        bn.load_state_dict(sd)
        xd = x.clone().requires_grad_(True)
        w.grad = bias.grad = None
        fork = o.GradFork()
        conv = Holder.__new__(Holder)
        conv.weight, conv.bias = w, bias
        spec = o.ConvSpec(3, 3, 1, 1, o.ACT_GELU if with_bn else o.ACT_NONE, bn if with_bn else None)
        y = o.conv_bn_act(xd, None, conv, bn if with_bn else None, spec, fork if C == N else None)
        paths = [lib.dm_last_conv_path(), lib.dm_last_conv_persistent()]
        loss = (y.float() * probe.float()).sum()
        if C == N:                                             # x feeds a second consumer: its gradient joins in the dgrad epilogue (addend)
            loss = loss + ((fork.second(xd).float() + 0.0 * y.float()) * 0.5).sum()   # (consumes y too: its backward runs before the conv's)
        loss.backward()
        return [y.detach().clone(), xd.grad.clone(), w.grad.clone(), bias.grad.clone(), bn.running_var.clone()], paths
    try:
        for with_bn in (False, True):
            a, pa = run(1, with_bn)
            b_, pb = run(0, with_bn)
            assert pa == [1, 1] and pb == [1, 0], (pa, pb)             # the forward launch took the persistent / the per-tile halo kernel
            for name, u, v in zip(("y", "dx", "dw", "db", "running_var"), a, b_):
                if name in ("y", "dx") and not with_bn:
                    assert torch.equal(u, v), (with_bn, name, float((u.float() - v.float()).abs().max()))
                else:    # BatchNorm statistics arrive through fp64 atomics (order-dependent in the last bits); dw through the split reduction
                    assert torch.allclose(u.float(), v.float(), rtol=2e-3, atol=2e-3 * float(v.float().abs().max())), (with_bn, name)
    finally:
        lib.dm_set_conv_persist(1)


def test_persistent_halo_kernel_exact_integers():
    o = ops()
    B, C, N, H = 52, 128, 128, 64                               # 832 tiles on 256 CUs: workgroups with three and with four tiles
    g = torch.Generator().manual_seed(77)
    ri = lambda *s: torch.randint(-1, 2, s, generator=g).float()
    x, w, b, probe = ri(B, C, H, H), ri(N, C, 3, 3), ri(N), ri(B, N, H, H)
    w = w * (torch.rand(N, C, 3, 3, generator=g) < 0.2).float()
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, b, padding=1)
    assert yr.abs().max() < 256
    (yr * probe).sum().backward()
    conv = Holder(w, b)
    d1 = nhwc(x, torch.bfloat16).requires_grad_(True)
    y = o.conv_bn_act(d1, None, conv, None, o.ConvSpec(3, 3, 1, 1))
    assert torch.equal(nchw(y), yr.detach())
    (y.float() * nhwc(probe)).sum().backward()
    assert torch.equal(nchw(d1.grad), xr.grad)
    assert torch.equal(conv.weight.grad.cpu(), wr.grad)


# ---- packed-tap kernel (igemm_skinny.hip): 3x3 layers over an 8-channel input — the stem and the head's input gradient -----------------
@pytest.mark.parametrize("B,Co,H,dtype", [(3, 128, 64, torch.bfloat16), (2, 192, 64, torch.float16), (1, 128, 128, torch.bfloat16), (5, 40, 64, torch.bfloat16),
                                          (2, 256, 64, torch.bfloat16)])      # (256: two channel tiles / two wide blocks, the cfg-5 stem)
def test_packed_tap_kernel_stem_forward_exact_integers(B, Co, H, dtype):
    """K = 9 taps x 8 channels packed into three MFMA k-steps (the fourth lane group of the last step multiplies zero weights with the
    pixel of tap 8): integer data must reproduce F.conv2d bit for bit, image borders, column tiles (H = 128) and a ragged second
    channel tile included; the weight gradient of the same layer still comes from the halo weight-gradient kernel."""
    o = ops()
    lib = __import__("diffusionmodel_amd")._lib.load()
    g = torch.Generator().manual_seed(B * 7 + Co + H)
    ri = lambda *s: torch.randint(-2, 3, s, generator=g).float()
    x, w, b, probe = ri(B, 8, H, H), ri(Co, 8, 3, 3), ri(Co), torch.randint(-1, 2, (B, Co, H, H), generator=g).float()
    probe = probe * (torch.rand(B, Co, H, H, generator=g) < 0.3).float()      # keeps the 8-channel input gradient below 256 (exact in bf16)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=1)
    assert yr.abs().max() < 256
    (yr * probe).sum().backward()
    assert xr.grad.abs().max() < 256
    conv = Holder(w, b)
    d1 = nhwc(x, dtype).requires_grad_(True)
    y = o.conv_bn_act(d1, None, conv, None, o.ConvSpec(3, 3, 1, 1))
    assert lib.dm_last_conv_path() == 4                       # the forward launch took the packed-tap kernel
    assert torch.equal(nchw(y), yr.detach())
    (y.float() * nhwc(probe)).sum().backward()
    assert torch.equal(nchw(d1.grad), xr.grad)                # (8 output channels: the gather kernel's 32-wide tiles)
    assert lib.dm_last_wgrad_path() == (4 if Co % 128 == 0 else 1)      # whole 128-channel blocks: wgrad3x3_skinny_kernel (stem form)
    assert torch.equal(conv.weight.grad.cpu(), wr.grad) and torch.equal(conv.bias.grad.cpu(), br.grad)


@pytest.mark.parametrize("B,C,H,dtype", [(3, 128, 64, torch.bfloat16), (2, 256, 64, torch.float16), (1, 128, 128, torch.bfloat16)])
def test_packed_tap_kernel_head_input_gradient_exact_integers(B, C, H, dtype):
    """The head (C -> 3 channels): the forward runs on conv3x3_narrow_kernel (input halo and all nine taps of 16 output channels
    resident, one image row per wave); its input gradient is a 3x3 convolution of the 8-channel (padded) output gradient with the
    transposed weights and mirrored taps — the packed-tap kernel's FLIP form."""
    o = ops()
    lib = __import__("diffusionmodel_amd")._lib.load()
    g = torch.Generator().manual_seed(B + C + H)
    ri = lambda *s: torch.randint(-1, 2, s, generator=g).float()
    x, w, b, probe = ri(B, C, H, H), ri(3, C, 3, 3), ri(3), ri(B, 3, H, H)
    w = w * (torch.rand(3, C, 3, 3, generator=g) < 0.2).float()
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=1)
    assert yr.abs().max() < 256
    (yr * probe).sum().backward()
    conv = Holder(w, b)
    d1 = nhwc(x, dtype).requires_grad_(True)
    y = o.conv_bn_act(d1, None, conv, None, o.ConvSpec(3, 3, 1, 1, out_nchw=True))      # the head's own call (modules.py: fp32 NCHW output)
    assert lib.dm_last_conv_path() == 5                       # forward: the narrow halo kernel (<= 16 output channels)
    assert torch.equal(y.float().cpu(), yr.detach())
    (y.float() * probe.to(DEV)).sum().backward()
    assert lib.dm_last_conv_path() == 4                       # the input-gradient launch (the only dm_conv of the backward pass)
    assert lib.dm_last_wgrad_path() == 4                      # wgrad3x3_skinny_kernel (head form: dy is the 8-channel side)
    assert torch.equal(nchw(d1.grad), xr.grad)
    assert torch.equal(conv.weight.grad.cpu(), wr.grad) and torch.equal(conv.bias.grad.cpu(), br.grad)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_packed_tap_kernel_matches_the_halo_kernel_with_batchnorm_and_gelu(dtype):
    """Train-mode stem (conv -> BatchNorm statistics in the epilogue -> GELU): same epilogue code as the halo kernel, another
    accumulation order in the k loop — outputs, running statistics and all gradients agree to rounding."""
    o = ops()
    from diffusionmodel_amd import _lib as L
    lib = L.load()
    torch.manual_seed(21)
    B, H, N = 6, 64, 128
    x = torch.randn(B, H, H, 8).to(dtype).to(DEV)
    x[..., 3:] = 0                                              # (the stem's five padding channels)
    w = torch.nn.Parameter((torch.randn(N, 8, 3, 3) / math.sqrt(27)).to(DEV).contiguous(memory_format=torch.channels_last))
    bias = torch.nn.Parameter(torch.randn(N).to(DEV) * 0.1)
    probe = torch.randn(B, H, H, N).to(dtype).to(DEV)
    bn = torch.nn.BatchNorm2d(N).to(DEV).train()
    sd = {k: v.clone() for k, v in bn.state_dict().items()}

    def run(on):
        lib.dm_set_conv_packtap(on)
        bn.load_state_dict(sd)
        w.grad = bias.grad = bn.weight.grad = bn.bias.grad = None
        conv = Holder.__new__(Holder)
        conv.weight, conv.bias = w, bias
        y = o.conv_bn_act(x, None, conv, bn, o.ConvSpec(3, 3, 1, 1, o.ACT_GELU, bn))
        path = lib.dm_last_conv_path()
        (y.float() * probe.float()).sum().backward()
        return [y.detach().float(), w.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_mean.clone(), bn.running_var.clone()], path
    try:
        a, pa = run(1)
        b_, pb = run(0)
    finally:
        lib.dm_set_conv_packtap(1)
    assert pa == 4 and pb == 1, (pa, pb)
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    for name, u, v in zip(("y", "dw", "dgamma", "dbeta", "running_mean", "running_var"), a, b_):
        tol = 2 * ulp if name == "y" else 2e-3
        assert float((u - v).abs().max()) <= tol * float(v.abs().max()), (name, float((u - v).abs().max()), float(v.abs().max()))


def test_lds_residue_audit_exact_integer_kernels_with_every_cu_lds_poisoned():
    """VERDICT r03 #8 (the r02 corruption, by audit instead of repetition): no workgroup is preempted mid-kernel on this stack, so what a
    foreign process can leave behind is what survives a kernel boundary on a CU — LDS above all.  A kernel that reads LDS bytes it did
    not write itself computes on its own benign residue when alone and on a stranger's data in company.  Deterministic check, one
    process, one pass: before EVERY library launch all 160 KiB of every CU's LDS are filled with a NaN bit pattern (fp32 NaN = two bf16
    NaNs = two fp16 NaNs: dm_debug_poison_lds, _lib.POISON_LDS), and the exact-integer tests of the LDS-heavy kernel families are run
    again: the halo convolution / its input gradient / the halo weight gradient (+ reduce), the four-tap halo kernels of the 4x4 stride-2
    layer (conv + weight gradient), the pointwise conv and weight-gradient kernels, the packed-tap / narrow / skinny kernels, the
    persistent halo kernel, the split-K last-arriver fold, the dense kernels.  Any residue read shows up as NaN in a bit-exact result."""
    from diffusionmodel_amd import _lib as L
    lib = L.load()
    L.POISON_CALLS[0] = 0
    L.POISON_LDS[0] = 0x7FC17FC1
    try:
        test_wgrad_bf16_exact_integers()
        assert lib.dm_set_conv_variant(5) == 0
        for case in HALO_CASES:
            test_conv_halo_kernel_exact_integers(case, None)
        for case in TAP4_CASES:
            for dtype in (torch.bfloat16, torch.float16):
                test_conv4x4s2_four_tap_halo_kernel_exact_integers(case, dtype)
        test_splitk_last_arriver_form_is_bit_exact_too()
        for dtype in (torch.bfloat16, torch.float16):
            for shp in [(2, 32, 128, 16), (1, 64, 64, 12), (2, 128, 32, 8), (3, 128, 256, 10), (1, 32, 40, 9), (2, 64, 136, 16)]:
                test_pointwise_conv_kernel_exact_integers(dtype, *shp)
        for dtype in (torch.bfloat16, torch.float16):
            for shp in [(2, 32, 128, 16), (3, 128, 32, 10), (4, 32, 256, 32), (1, 256, 32, 9), (2, 64, 512, 16), (3, 256, 64, 8), (2, 128, 128, 16),
                        (1, 128, 384, 12), (3, 512, 128, 8)]:
                test_pointwise_wgrad_kernel_exact_integers(dtype, *shp)
        test_persistent_halo_kernel_exact_integers()
        for args in [(3, 128, 64, torch.bfloat16), (2, 192, 64, torch.float16), (1, 128, 128, torch.bfloat16)]:
            test_packed_tap_kernel_stem_forward_exact_integers(*args)
        for args in [(3, 128, 64, torch.bfloat16), (2, 256, 64, torch.float16), (1, 128, 128, torch.bfloat16)]:
            test_packed_tap_kernel_head_input_gradient_exact_integers(*args)
        for mkn in [(64, 1024, 1024), (64, 1024, 64), (2048, 128, 8)]:
            test_dense_layer_kernels_exact_integers(*mkn)
    finally:
        L.POISON_LDS[0] = None
        lib.dm_set_conv_variant(L.DEFAULT_CONV_VARIANT)
    assert L.POISON_CALLS[0] > 500, L.POISON_CALLS[0]          # the poison launches really ran


def test_coordattn_chain_respects_the_64_kib_rule_on_a_shared_gpu():
    """when the device guard has switched the library to its <= 64-KiB-LDS kernels (conv variant 2: a GPU shared between
    processes), the fused CoordAttn chain must not launch workgroups above 64 KiB either.  Strips of 128 + 128 positions x 16 channels need
    61.9 KiB of dynamic LDS (+ 22 KiB static): the C ABI refuses them (DM_EUNSUPPORTED), ops.ca_chain_ok routes the block to the
    one-launch-per-op path, and that path's result equals the chain's (fp32)."""
    import diffusionmodel_amd as D
    from diffusionmodel_amd import _lib as L, ops as o
    lib = L.load()
    torch.manual_seed(3)
    ca = D.CoordAttn(256).to(DEV).eval()
    with torch.no_grad():
        for p_ in ca.parameters():
            p_.normal_(0, 0.2)
    x = torch.randn(2, 256, 128, 128, device=DEV)
    assert o.ca_chain_lds(128, 128, 16) + 22 * 1024 > 64 * 1024
    try:
        with torch.no_grad():
            assert o.ca_chain_ok(256, 16, 128, 128)
            y_chain = ca(x)
            assert lib.dm_set_conv_variant(2) == 0
            assert not o.ca_chain_ok(256, 16, 128, 128) and o.ca_chain_ok(256, 16, 32, 32)
            y_small = ca(x)                                   # per-op path (H == W)
            d = L.DmCaChain()
            d.B, d.H, d.W, d.C, d.R = 2, 128, 128, 256, 16
            import ctypes as C
            assert lib.dm_ca_chain_fwd(C.byref(d), None) == -2 and b"64-KiB" in lib.dm_last_error()
    finally:
        lib.dm_set_conv_variant(L.DEFAULT_CONV_VARIANT)
    assert rel_err(y_small.cpu(), y_chain.cpu()) < 2e-4


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("B,H,C,N,dtype", [(6, 64, 128, 128, torch.bfloat16), (3, 64, 64, 192, torch.float16), (2, 128, 64, 128, torch.bfloat16)])
def test_four_wave_halo_kernel_is_bit_identical_to_the_eight_wave_kernel(B, H, C, N, dtype, mode):
    """r04 experiment (igemm_halo4.hip, dm_set_conv_wave4): the 3x3 halo kernel with one wave per SIMD — four waves, 128 x 64 wave tiles,
    the shared epilogue once per image row of a wave — keeps the tile, the LDS image and the order of the fp32 sums of the eight-wave
    kernel: forward (plain and with BatchNorm statistics + GELU), input gradient with the forked-gradient addend, ragged N, 128-pixel
    rows (column tiles) must agree BIT FOR BIT on random data (statistics to the fp64-atomics band)."""
    o = ops()
    from diffusionmodel_amd import _lib as L
    lib = L.load()
    torch.manual_seed(B + H + C + N)
    x = torch.randn(B, H, H, C).to(dtype).to(DEV)
    w = torch.nn.Parameter((torch.randn(N, C, 3, 3) / math.sqrt(9 * C)).to(DEV).contiguous(memory_format=torch.channels_last))
    bias = torch.nn.Parameter(torch.randn(N).to(DEV) * 0.1)
    probe = torch.randn(B, H, H, N).to(dtype).to(DEV)
    bn = torch.nn.BatchNorm2d(N).to(DEV)
    bn.train()
    sd = {k: v.clone() for k, v in bn.state_dict().items()}

    def run(wave4, with_bn):
        lib.dm_set_conv_wave4(wave4)
        bn.load_state_dict(sd)
        xd = x.clone().requires_grad_(True)
        w.grad = bias.grad = None
        fork = o.GradFork()
        conv = Holder.__new__(Holder)
        conv.weight, conv.bias = w, bias
        spec = o.ConvSpec(3, 3, 1, 1, o.ACT_GELU if with_bn else o.ACT_NONE, bn if with_bn else None)
        y = o.conv_bn_act(xd, None, conv, bn if with_bn else None, spec, fork if C == N else None)
        loss = (y.float() * probe.float()).sum()
        if C == N:
            loss = loss + ((fork.second(xd).float() + 0.0 * y.float()) * 0.5).sum()
        loss.backward()
        return [y.detach().clone(), xd.grad.clone(), bn.running_var.clone(), bn.running_mean.clone()]
    try:
        lib.dm_set_conv_persist(0)
        for with_bn in (False, True):
            a = run(mode, with_bn)
            b_ = run(0, with_bn)
            for name, u, v in zip(("y", "dx", "running_var", "running_mean"), a, b_):
                if name in ("y", "dx") and not with_bn:
                    assert torch.equal(u, v), (with_bn, name, float((u.float() - v.float()).abs().max()))
                else:
                    assert torch.allclose(u.float(), v.float(), rtol=2e-3, atol=2e-3 * float(v.float().abs().max())), (with_bn, name)
    finally:
        lib.dm_set_conv_wave4(0)
        lib.dm_set_conv_persist(1)
