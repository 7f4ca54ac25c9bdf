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
    monkeypatch.setattr(ns, "gen_samples", lambda ckpt, n_samples_per_class, guide_scales, eval_quality, data_root:
                        calls.update(ckpt=ckpt, n=n_samples_per_class, scales=guide_scales, ev=eval_quality))
    monkeypatch.setattr(ns, "train_model", lambda max_epochs=None, data_root=None: calls.update(train=max_epochs))
    ns.main(["--mode", "generate", "--ckpt", "a.pt", "--guide_scales", "2", "4", "--samples", "5", "--no_eval"])
    assert calls == {"ckpt": "a.pt", "n": 5, "scales": [2.0, 4.0], "ev": False}
    calls.clear()
    ns.main(["--mode", "generate", "--checkpoint", "b.pt", "--guidance_scales", "6", "--samples_per_class", "2", "--no_memory_cleanup"])
    assert calls == {"ckpt": "b.pt", "n": 2, "scales": [6.0], "ev": True}
    calls.clear()
    ns.main(["--mode", "train", "--epochs", "3"])
    assert calls == {"train": 3}
    with pytest.raises(SystemExit):
        ns.main(["--mode", "generate"])


def test_early_stop_patience():
    """new_scripy.py:587-620: the call returns "improved" (the caller writes best_model.pt), `.early_stop` flips after `patience`
    epochs without an improvement of at least min_delta, best_state is the 3-key dict of best_model_early.pt."""
    import new_scripy as ns
    es = ns.EarlyStop(patience=2, min_delta=0.1, verbose=False)
    assert es(1.0, None, 0) is True and es.best_state["epoch"] == 0 and set(es.best_state) == {"epoch", "model_state_dict", "val_loss"}
    assert es(0.95, None, 1) is False and es.counter == 1 and not es.early_stop      # not better by min_delta
    assert es(0.96, None, 2) is False and es.early_stop                              # counter 2 -> stop
    es = ns.EarlyStop(patience=2, min_delta=0.1, verbose=False)
    es(1.0, None, 0)
    assert es(0.8, None, 1) is True
    assert es.counter == 0 and es.best_loss == 0.8 and es.best_state["val_loss"] == 0.8


def test_grid_and_image_writers(tmp_path):
    """save_samples = make_grid(nrow, padding=2) + save_image; save_image quantises like torchvision (x*255+0.5)."""
    import numpy as np
    from PIL import Image
    import new_scripy as ns
    x = torch.linspace(-1, 1, 6 * 3 * 4 * 5).reshape(6, 3, 4, 5)
    ns.save_samples(x, str(tmp_path / "g.png"), nrow=4)
    g = np.array(Image.open(tmp_path / "g.png"))
    assert g.shape == (2 * (4 + 2) + 2, 4 * (5 + 2) + 2, 3)                # 2 rows of 4 with 2-pixel padding
    assert (g[:2] == 0).all() and (g[:, :2] == 0).all()
    want = ((x[5] * 0.5 + 0.5).clamp(0, 1) * 255 + 0.5).to(torch.uint8).permute(1, 2, 0).numpy()
    assert (g[2 + 6:2 + 6 + 4, 2 + 7:2 + 7 + 5] == want).all()             # image 5 sits in row 1, column 1
    ns.save_image(x[0], str(tmp_path / "i.png"))
    assert np.array(Image.open(tmp_path / "i.png")).shape == (4, 5, 3)


def test_generate_infers_class_count_from_checkpoint():
    import new_scripy as ns
    assert ns._classes_of_checkpoint({"nn_model.ctx_emb1.model.0.weight": torch.zeros(256, 5)}) == 5
    assert ns._classes_of_checkpoint({"ctx_emb1.model.0.weight": torch.zeros(256, 7)}) == 7
    with pytest.raises(KeyError):
        ns._classes_of_checkpoint({"foo": torch.zeros(1)})


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
        ddpm, hist = ns.train_model(n_classes=4, n_train=18, n_val=8, max_epochs=2, quiet=True)       # 18 = 4 full batches + a tail of 2
        assert len(hist) == 2 and all(torch.isfinite(torch.tensor(h["train_loss"])) for h in hist)
        ck = os.path.join(Cfg.SAVE_DIR, "ckpt_ep1.pt")
        assert os.path.isfile(ck)
        sd = torch.load(ck, map_location="cpu", weights_only=True)
        assert set(sd) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "loss", "metrics"}
        # the optimiser state is torch.optim.AdamW's schema (new_scripy.py:736-743 saves optim.state_dict())
        osd = sd["optimizer_state_dict"]
        n_par = len(list(ddpm.parameters()))
        assert osd["param_groups"][0]["params"] == list(range(n_par)) and len(osd["state"]) == n_par
        assert set(osd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
        # 5 micro-batches per epoch with ACCUM_STEPS = 2 -> 3 optimiser steps per epoch (the tail group is flushed, :795)
        assert float(osd["state"][0]["step"]) == 6.0
        assert set(sd["metrics"]) == {"train_loss", "val_loss", "img_metrics", "lr"} and len(sd["metrics"]["train_loss"]) == 2
        best = torch.load(os.path.join(Cfg.SAVE_DIR, "best_model.pt"), map_location="cpu", weights_only=True)
        assert set(best) == set(sd)                                                   # the 6-key dict (:898-901)
        assert os.path.isfile(os.path.join(Cfg.SAVE_DIR, "metrics", "metrics_ep1.json"))
        assert os.path.isfile(os.path.join(Cfg.SAVE_DIR, "img_ep0_w2.0.png"))         # periodic sampling at ep % 5 == 0 (:851)
        assert sd["metrics"]["img_metrics"] and {"ssim", "psnr", "guide_scale", "epoch"} <= set(sd["metrics"]["img_metrics"][0])
        real = torch.rand(4, 3, 64, 64) * 2 - 1
        out = ns.gen_samples(ck, n_samples_per_class=1, guide_scales=[2.0], real_images=real)          # class count from the checkpoint
        x = out[2.0]["samples"]
        assert x.shape == (4, 3, 64, 64) and torch.isfinite(x).all()
        d = os.path.dirname(out[2.0]["grid_path"])
        assert os.path.isfile(out[2.0]["grid_path"]) and os.path.basename(out[2.0]["grid_path"]) == "samples_g2.0.png"
        assert all(os.path.isfile(os.path.join(d, f"class{i}_s0_g2.0.png")) for i in range(4))
        assert os.path.isfile(os.path.join(d, "quality_metrics.json"))
        # resume: a fresh optimiser takes the saved state back (moments, step) in torch's layout
        import diffusionmodel_amd as D
        d2 = ns.build_model(4, "cuda:0")
        d2.load_state_dict(sd["model_state_dict"])
        o2 = D.FusedAdamW(d2.parameters(), lr=Cfg.LR, weight_decay=Cfg.WD)
        o2.load_state_dict(osd)
        assert o2._step == 6 and int(o2._step_dev.item()) == 6
        back = o2.state_dict()
        for i in (0, 7, n_par - 1):
            assert torch.equal(back["state"][i]["exp_avg"].cpu(), osd["state"][i]["exp_avg"])
            assert torch.equal(back["state"][i]["exp_avg_sq"].cpu(), osd["state"][i]["exp_avg_sq"])
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
        # --mode generate with the dataset at hand: class names and the real images of the quality pass come from it (new_scripy.py:960-961, 1001-1029)
        out = ns.gen_samples(os.path.join(Cfg.SAVE_DIR, "ckpt_ep0.pt"), n_samples_per_class=2, guide_scales=[2.0], data_root=root)
        d = os.path.dirname(out[2.0]["grid_path"])
        assert out[2.0]["samples"].shape == (4, 3, 64, 64)
        assert all(os.path.isfile(os.path.join(d, f"{cls}_s{k}_g2.0.png")) for cls in ("crack_a", "crack_b") for k in (0, 1))
        import json
        q = json.load(open(os.path.join(d, "quality_metrics.json")))
        assert set(q["2.0"]) >= {"ssim", "psnr"}
    finally:
        for k, v in saved.items():
            setattr(Cfg, k, v)


@pytest.mark.gpu
def test_mnist_script_train_and_sample_synthetic(tmp_path):
    """BASELINE configs[0] / SURVEY §2 row 6: `train_mnist` (MNIST_script.py:303-394) restated on synthetic digits — a short run
    trains (the loss EMA falls), samples at every guidance weight and writes the per-epoch grids."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import MNIST_script as ms
    imgs, labels = ms.synthetic_digits(96, 10, seed=1)
    assert imgs.shape == (96, 1, 28, 28) and float(imgs.min()) >= 0 and float(imgs.max()) <= 1
    ddpm, hist = ms.train_mnist(n_epoch=3, batch_size=32, n_T=6, n_feat=32, lrate=2e-3, save_dir=str(tmp_path) + "/", ws_test=(0.0, 2.0),
                                data=(imgs, labels), quiet=True, save_model=True)
    assert len(hist) == 3 and all(h == h for h in hist) and hist[-1] < hist[0]
    for ep in range(3):
        for w in (0.0, 2.0):
            assert os.path.isfile(os.path.join(str(tmp_path), f"image_ep{ep}_w{w}.png"))
    sd = torch.load(os.path.join(str(tmp_path), "model_2.pth"), map_location="cpu", weights_only=True)
    assert "nn_model.init_conv.conv1.0.weight" in sd and "sqrtab" in sd


@pytest.mark.gpu
def test_two_rank_cli_train_early_stop_then_generate_end_to_end(tmp_path):
    """the multi-rank path of the DRIVER itself (new_scripy.py:777-848, 1036-1061 on diffusionmodel_amd/train.py),
    through the command line, as two gloo ranks on one GPU (DM_DIST_BACKEND=gloo; RCCL refuses two ranks per device):
    `--mode train --gpus 2` on 20 synthetic images in micro-batches of 4 = FIVE micro-batches per epoch (rank 1 idles in the tail
    group), ACCUM_STEPS 2, an early stop in the second epoch (patience 1, min_delta 10) — every rank must leave (no rank waits in a
    collective), only rank 0 writes, the ranks' parameters are identical at exit, the checkpoint loads in a single process; then
    `--mode generate --gpus 2 --seed 11` == the single-process `--mode generate --seed 11` of the same checkpoint."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import subprocess
    env = dict(os.environ, DM_DIST_BACKEND="gloo", DM_DEVICE_GUARD="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    save, samp = str(tmp_path / "ckpt") + "/", str(tmp_path / "samples") + "/"
    size = ["--img_size", "64", "--n_feat", "32", "--n_T", "8", "--batch_size", "4", "--bottleneck_k", "4", "--dtype", "float32",
            "--save_dir", save, "--sample_dir", samp]
    script = os.path.join(ROOT, "new_scripy.py")
    r = subprocess.run([sys.executable, script, "--mode", "train", "--gpus", "2", "--epochs", "4", "--accum_steps", "2", "--patience", "1",
                        "--min_delta", "10", "--n_train", "20", "--n_val", "8", "--guide_scales", "2"] + size,
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = r.stdout                                  # rank 0's stdout; rank 1's goes to stderr (parallel.launch_ranks)
    assert "epoch 0:" in out and "epoch 1:" in out and "epoch 2:" not in out, out[-2000:]           # stopped in the second epoch
    assert "Early stopping triggered" in out and "parameters identical on 2 ranks" in out, out[-2000:]
    assert "Saved" not in r.stderr and "epoch 0:" not in r.stderr                                    # rank 1 wrote and said nothing
    assert out.count("Saved best checkpoint") == 1 and os.path.isfile(save + "best_model.pt") and os.path.isfile(save + "best_model_early.pt")
    assert os.path.isfile(save + "img_ep0_w2.0.png") and os.path.isfile(save + "metrics/metrics_ep0.json")
    final = save + "ckpt_ep3.pt"                                                                     # the final save of :931
    assert os.path.isfile(final)
    ck = torch.load(final, map_location="cpu", weights_only=True)
    # 5 micro-batches per epoch, G = 2: steps cover (0,1) (2,3) (4,idle) -> 3 optimiser steps per epoch, 2 epochs
    assert float(ck["optimizer_state_dict"]["state"][0]["step"]) == 6.0 and len(ck["metrics"]["train_loss"]) == 2
    import new_scripy as ns
    from diffusionmodel_amd import Cfg
    saved = {k: getattr(Cfg, k) for k in ("IMG_SIZE", "N_FEAT", "N_T", "BOTTLENECK_K", "DTYPE")}
    try:
        Cfg.IMG_SIZE, Cfg.N_FEAT, Cfg.N_T, Cfg.BOTTLENECK_K, Cfg.DTYPE = 64, 32, 8, 4, "float32"
        d1 = ns.build_model(4, "cuda:0")
        d1.load_state_dict(ck["model_state_dict"])                                                   # loads single-process
    finally:
        for k, v in saved.items():
            setattr(Cfg, k, v)
    imgs = {}
    for tag, extra in (("two", ["--gpus", "2"]), ("one", [])):
        sdir = str(tmp_path / ("gen_" + tag)) + "/"
        g = subprocess.run([sys.executable, script, "--mode", "generate", "--ckpt", final, "--samples", "2", "--guide_scales", "2", "--no_eval",
                            "--seed", "11", "--save_raw"] + extra + size[:-1] + [sdir],
                           capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert g.returncode == 0, g.stdout[-2000:] + g.stderr[-3000:]
        runs = [d for d in os.listdir(sdir) if d.startswith("samples_")]
        assert len(runs) == 1 and "guide scale 2.0: wrote" in g.stdout and "wrote" not in g.stderr
        d = os.path.join(sdir, runs[0])
        assert os.path.isfile(os.path.join(d, "samples_g2.0.png")) and all(os.path.isfile(os.path.join(d, f"class{i}_s{k}_g2.0.png")) for i in range(4) for k in (0, 1))
        imgs[tag] = torch.load(os.path.join(d, "samples_g2.0.pt"), weights_only=True)
    assert imgs["two"].shape == (8, 3, 64, 64) and torch.isfinite(imgs["two"]).all()
    err = (imgs["two"] - imgs["one"]).abs().max().item()
    assert err <= 2e-5 * 8, err                     # 8 steps; per step only the split-K summation order of a few layers follows the batch size
