"""§8f rows 3-4: global-statistics SSIM / PSNR and the CrackDataset attention mask — oracle vs the reference-generated
fixture (CPU) and the HIP kernels vs both (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_ref as MR
from oracle import synth

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics.npz"))


def pairs():
    out = []
    for i in range(6):
        a = synth.synth_input(f"metrics.a{i}", (3, 32, 32))
        b = a + 0.2 * synth.synth_input(f"metrics.b{i}", (3, 32, 32))
        if i % 2 == 1:
            a, b = (a.clamp(-1, 1) + 1) / 2, (b.clamp(-1, 1) + 1) / 2
        if i == 4:
            b = (b.clamp(-1, 1) + 1) / 2
        out.append((a, b))
    return out


def test_oracle_metrics_match_reference():
    for i, (a, b) in enumerate(pairs()):
        assert abs(MR.calc_ssim(a, b) - G["ssim"][i]) < 1e-6
        assert abs(MR.calc_psnr(a, b) - G["psnr"][i]) < 1e-5
    p = pairs()
    ev = MR.evaluate_batch(torch.stack([q[0] for q in p[:4]]), torch.stack([q[1] for q in p[:4]]))
    assert abs(ev["ssim"] - float(G["eval.ssim"])) < 1e-6 and abs(ev["psnr"] - float(G["eval.psnr"])) < 1e-5


def test_oracle_masks_match_reference():
    for S in (64, 256):
        for j, (x0, y0, x1, y1, w, h) in enumerate(G["boxes"]):
            m = MR.attn_mask(MR.scaled_bbox(int(x0), int(y0), int(x1), int(y1), int(w), int(h), S), S)
            assert np.array_equal(m.numpy(), G[f"mask.S{S}.{j}"]), (S, j)


def test_scaled_bbox_host_logic_matches_oracle():
    from diffusionmodel_amd import data
    for (x0, y0, x1, y1, w, h) in G["boxes"]:
        for S in (64, 256):
            assert data.scaled_bbox(int(x0), int(y0), int(x1), int(y1), int(w), int(h), S) == \
                MR.scaled_bbox(int(x0), int(y0), int(x1), int(y1), int(w), int(h), S)


@pytest.mark.gpu
def test_hip_metrics_and_masks():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from diffusionmodel_amd import data
    from diffusionmodel_amd.metrics import ImageMetrics
    for i, (a, b) in enumerate(pairs()):
        assert abs(ImageMetrics.calc_ssim(a.cuda(), b.cuda()) - G["ssim"][i]) < 2e-6
        assert abs(ImageMetrics.calc_psnr(a.cuda(), b.cuda()) - G["psnr"][i]) < 2e-5
    p = pairs()
    ev = ImageMetrics().evaluate_batch(torch.stack([q[0] for q in p[:4]]).cuda(), torch.stack([q[1] for q in p[:4]]).cuda())
    assert abs(ev["ssim"] - float(G["eval.ssim"])) < 2e-6 and abs(ev["psnr"] - float(G["eval.psnr"])) < 2e-5
    x = torch.rand(3, 16, 16).cuda()
    assert ImageMetrics.calc_psnr(x, x.clone()) == float("inf")
    for S in (64, 256):
        boxes = [data.scaled_bbox(int(x0), int(y0), int(x1), int(y1), int(w), int(h), S) for (x0, y0, x1, y1, w, h) in G["boxes"]]
        m = data.attn_masks(boxes, S).cpu().numpy()
        for j in range(len(boxes)):
            assert np.array_equal(m[j], G[f"mask.S{S}.{j}"]), (S, j)
