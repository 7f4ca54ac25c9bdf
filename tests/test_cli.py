"""`new_scripy.py --mode train|generate` drop-in: flag spellings (CPU) and an end-to-end synthetic smoke (GPU)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cli_accepts_both_flag_spellings(monkeypatch):
    import new_scripy as ns
    calls = {}
    monkeypatch.setattr(ns, "gen_samples", lambda ckpt, n, scales: calls.update(ckpt=ckpt, n=n, scales=scales))
    monkeypatch.setattr(ns, "train_model", lambda max_epochs=None, data_root=None: calls.update(train=max_epochs))
    ns.main(["--mode", "generate", "--ckpt", "a.pt", "--guide_scales", "2", "4", "--samples", "5", "--no_eval"])
    assert calls == {"ckpt": "a.pt", "n": 5, "scales": [2.0, 4.0]}
    calls.clear()
    ns.main(["--mode", "generate", "--checkpoint", "b.pt", "--guidance_scales", "6", "--samples_per_class", "2", "--no_memory_cleanup"])
    assert calls == {"ckpt": "b.pt", "n": 2, "scales": [6.0]}
    calls.clear()
    ns.main(["--mode", "train", "--epochs", "3"])
    assert calls == {"train": 3}
    with pytest.raises(SystemExit):
        ns.main(["--mode", "generate"])


def test_early_stop_patience():
    import new_scripy as ns
    es = ns.EarlyStop(patience=2, min_delta=0.1)
    assert not es(1.0, None, 0)
    assert not es(0.95, None, 1)      # not better by min_delta -> counter 1
    assert es(0.96, None, 2)          # counter 2 -> stop
    es = ns.EarlyStop(patience=2, min_delta=0.1)
    es(1.0, None, 0)
    es(0.8, None, 1)
    assert es.counter == 0 and es.best_loss == 0.8


@pytest.mark.gpu
def test_train_then_generate_synthetic(tmp_path, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import new_scripy as ns
    from diffusionmodel_amd import Cfg
    saved = {k: getattr(Cfg, k) for k in ("IMG_SIZE", "N_FEAT", "N_T", "BATCH_SIZE", "ACCUM_STEPS", "BOTTLENECK_K", "SAVE_DIR", "SAMPLE_DIR", "GUIDE_SCALES", "DTYPE")}
    try:
        Cfg.IMG_SIZE, Cfg.N_FEAT, Cfg.N_T, Cfg.BATCH_SIZE, Cfg.ACCUM_STEPS, Cfg.BOTTLENECK_K = 64, 32, 8, 4, 2, 4
        Cfg.SAVE_DIR, Cfg.SAMPLE_DIR, Cfg.GUIDE_SCALES = str(tmp_path / "ckpt") + "/", str(tmp_path / "samples") + "/", [2.0]
        ddpm, hist = ns.train_model(n_classes=4, n_train=16, n_val=8, max_epochs=2, quiet=True)
        assert len(hist) == 2 and all(torch.isfinite(torch.tensor(h["train_loss"])) for h in hist)
        ck = os.path.join(Cfg.SAVE_DIR, "ckpt_ep1.pt")
        assert os.path.isfile(ck)
        sd = torch.load(ck, map_location="cpu", weights_only=True)
        assert set(sd) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "loss", "metrics"}
        out = ns.gen_samples(ck, n_samples_per_class=1, guide_scales=[2.0], n_classes=4)
        assert out[2.0].shape == (4, 3, 64, 64) and torch.isfinite(out[2.0]).all()
        assert os.path.isfile(os.path.join(Cfg.SAMPLE_DIR, "generated_w2.0.png"))
    finally:
        for k, v in saved.items():
            setattr(Cfg, k, v)


@pytest.mark.gpu
def test_train_on_dataset_directory(tmp_path):
    """--data_root path: images + VOC-XML on disk -> CrackDataset -> boxes -> masks rasterised on the device -> train step."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image
    import new_scripy as ns
    from diffusionmodel_amd import Cfg
    root = str(tmp_path / "ds")
    g = torch.Generator().manual_seed(0)
    for ci, cls in enumerate(("crack_a", "crack_b")):
        os.makedirs(os.path.join(root, "images", cls), exist_ok=True)
        os.makedirs(os.path.join(root, "annotations"), exist_ok=True)
        for j in range(10):
            name = f"{cls}_{j}"
            arr = (torch.rand(48, 80, 3, generator=g) * 255).to(torch.uint8).numpy()
            Image.fromarray(arr).save(os.path.join(root, "images", cls, name + ".png"))
            with open(os.path.join(root, "annotations", name + ".xml"), "w") as f:
                f.write(f"<annotation><size><width>80</width><height>48</height></size><object><bndbox><xmin>{5 + j}</xmin>"
                        f"<ymin>{3 + ci}</ymin><xmax>{40 + j}</xmax><ymax>{30 + ci}</ymax></bndbox></object></annotation>")
    keys = ("IMG_SIZE", "N_FEAT", "N_T", "BATCH_SIZE", "ACCUM_STEPS", "BOTTLENECK_K", "SAVE_DIR", "SAMPLE_DIR", "GUIDE_SCALES", "NUM_WORKERS")
    saved = {k: getattr(Cfg, k) for k in keys}
    try:
        Cfg.IMG_SIZE, Cfg.N_FEAT, Cfg.N_T, Cfg.BATCH_SIZE, Cfg.ACCUM_STEPS, Cfg.BOTTLENECK_K, Cfg.NUM_WORKERS = 64, 32, 8, 4, 2, 4, 0
        Cfg.SAVE_DIR, Cfg.SAMPLE_DIR, Cfg.GUIDE_SCALES = str(tmp_path / "ckpt") + "/", str(tmp_path / "samples") + "/", [2.0]
        ddpm, hist = ns.train_model(max_epochs=1, quiet=True, data_root=root)
        assert ddpm.n_classes == 2 and len(hist) == 1
        assert torch.isfinite(torch.tensor(hist[0]["train_loss"])) and torch.isfinite(torch.tensor(hist[0]["val_loss"]))
    finally:
        for k, v in saved.items():
            setattr(Cfg, k, v)
