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


def _write_voc(root, cls, name, size_wh, box, color):
    from PIL import Image
    os.makedirs(os.path.join(root, "images", cls), exist_ok=True)
    os.makedirs(os.path.join(root, "annotations"), exist_ok=True)
    Image.new("RGB", size_wh, color).save(os.path.join(root, "images", cls, name + ".png"))
    w, h = size_wh
    with open(os.path.join(root, "annotations", name + ".xml"), "w") as f:
        f.write(f"<annotation><size><width>{w}</width><height>{h}</height></size><object><bndbox><xmin>{box[0]}</xmin>"
                f"<ymin>{box[1]}</ymin><xmax>{box[2]}</xmax><ymax>{box[3]}</ymax></bndbox></object></annotation>")


def test_crack_dataset_matches_the_reference_mask_recipe(tmp_path):
    """CrackDataset (reference layout, no torchvision): class order, normalisation, and the mask recipe of
    new_scripy.py:533-546 as restated by oracle/metrics_ref.attn_mask."""
    import torch
    from diffusionmodel_amd import Cfg
    from diffusionmodel_amd.data import CrackDataset
    from oracle import metrics_ref as R
    root = str(tmp_path / "ds")
    _write_voc(root, "pothole", "a", (200, 100), (20, 10, 120, 90), (255, 0, 0))
    _write_voc(root, "alligator", "b", (64, 64), (0, 0, 63, 63), (0, 255, 0))
    _write_voc(root, "pothole", "c", (96, 48), (90, 40, 95, 47), (0, 0, 255))
    ds = CrackDataset(root, img_size=32)
    assert ds.classes == ["alligator", "pothole"] and len(ds) == 3
    for i in range(len(ds)):
        x, label, mask = ds[i]
        img_path, xml_path, lab = ds.samples[i]
        assert label == lab and x.shape == (3, 32, 32) and mask.shape == (32, 32)
        assert float(x.min()) >= -1.0 and float(x.max()) <= 1.0
        xmin, ymin, xmax, ymax, w, h = CrackDataset.read_box(xml_path)
        ref = R.attn_mask(R.scaled_bbox(xmin, ymin, xmax, ymax, w, h, 32), 32, Cfg.LOW_WEIGHT, Cfg.MID_WEIGHT, Cfg.HIGH_WEIGHT)
        assert torch.equal(mask, torch.as_tensor(ref, dtype=torch.float32))
    x, _, _ = ds[1]                     # pothole/a, a pure red image: channels (1, -1, -1) after normalisation
    assert torch.allclose(x[0], torch.ones(32, 32)) and torch.allclose(x[1], -torch.ones(32, 32))
    xb, lb, box = CrackDataset(root, img_size=32, return_boxes=True)[2]
    assert box.dtype == torch.int32 and box.tolist() == list(R.scaled_bbox(90, 40, 95, 47, 96, 48, 32))


def test_convert_supervisely_layout(tmp_path):
    """DatasetNinja split (img/ + ann/*.json, the bundled road-damage format) -> the reference's dataset layout."""
    import json
    from PIL import Image
    from diffusionmodel_amd.data import CrackDataset, convert_supervisely
    src = tmp_path / "train"
    (src / "img").mkdir(parents=True)
    (src / "ann").mkdir()
    for name, cls, pts in (("x1.jpg", "alligator crack", [[30, 8], [5, 20]]), ("x2.jpg", "pothole", [[1, 2], [10, 12]])):
        Image.new("RGB", (40, 30), (9, 9, 9)).save(str(src / "img" / name))
        json.dump({"size": {"height": 30, "width": 40},
                   "objects": [{"geometryType": "rectangle", "classTitle": cls, "points": {"exterior": pts, "interior": []}}]},
                  open(str(src / "ann" / (name + ".json")), "w"))
    json.dump({"size": {"height": 30, "width": 40}, "objects": []}, open(str(src / "ann" / "empty.jpg.json"), "w"))
    dst = str(tmp_path / "voc")
    assert convert_supervisely(str(src), dst) == 2
    ds = CrackDataset(dst, img_size=16)
    assert ds.classes == ["alligator_crack", "pothole"] and len(ds) == 2
    assert CrackDataset.read_box(ds.samples[0][1]) == (5, 8, 30, 20, 40, 30)
