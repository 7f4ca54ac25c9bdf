#!/usr/bin/env python3
"""Drop-in for the reference's `new_scripy.py --mode train|generate` on MI355X.

The model / DDPM classes come from `diffusionmodel_amd` (HIP kernels); this file restates the two
drivers around them — `train_model()` (new_scripy.py:659-943: gradient accumulation, clip 1.0, AdamW,
CosineAnnealingWarmRestarts(10, 2, 3e-5) stepped per epoch, early stopping, checkpoint dict format,
periodic sampling) and `gen_samples()` (:945-1108: checkpoint load with raw-state-dict fallback,
class-cycled CFG sampling per guide scale, PNG grids) — and the CLI with BOTH flag spellings
(code: --ckpt --guide_scales --samples --no_eval; README: --checkpoint --guidance_scales
--samples_per_class --no_memory_cleanup).  Out of scope by design: the VOC-XML crack
dataset and FID/SSIM/PSNR; without `--data` a synthetic dataset with the reference's mask-value
convention (0.5 / 1.0 / 3.0, new_scripy.py:535-546) is used.
"""
import argparse
import json
import os
import time

import numpy as np
import torch

from diffusionmodel_amd import Cfg, ContextUnet, DDPM, FusedAdamW


class SyntheticCrackDataset(torch.utils.data.Dataset):
    """(img (3,S,S) in [-1,1], label, attn_mask (S,S)) with the mask convention of new_scripy.py:535-546."""

    def __init__(self, n, size, n_classes, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.classes = [f"class{i}" for i in range(n_classes)]
        self.labels = torch.randint(0, n_classes, (n,), generator=g)
        self.imgs = torch.randn(n, 3, size, size, generator=g).clamp_(-1, 1)
        self.masks = torch.full((n, size, size), 0.5)
        self.masks[:, size // 2:, :] = 1.0
        for i in range(n):
            y0, x0 = [int(v) for v in torch.randint(0, size // 2, (2,), generator=g)]
            self.masks[i, y0:y0 + size // 4, x0:x0 + size // 4] = 3.0

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, i):
        return self.imgs[i], int(self.labels[i]), self.masks[i]


class EarlyStop:
    """Patience / min-delta convergence heuristic (new_scripy.py:587-620)."""

    def __init__(self, patience=Cfg.PATIENCE, min_delta=Cfg.MIN_DELTA, save_path=None):
        self.patience, self.min_delta, self.save_path = patience, min_delta, save_path
        self.counter, self.best_loss, self.early_stop = 0, None, False

    def __call__(self, val_loss, model, epoch):
        if self.best_loss is None or val_loss < self.best_loss - self.min_delta:
            self.best_loss, self.counter = val_loss, 0
            if self.save_path:
                torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "val_loss": val_loss}, self.save_path)
        else:
            self.counter += 1
            if self.counter >= self.patience:
                self.early_stop = True
        return self.early_stop


def save_samples(x, path, nrow):
    """PNG grid without torchvision (new_scripy.py:554-561 used save_image/make_grid). x in [-1,1]."""
    from PIL import Image
    x = ((x.detach().float().cpu().clamp(-1, 1) + 1) * 127.5).round().to(torch.uint8)
    n, c, h, w = x.shape
    ncol = nrow
    nr = (n + ncol - 1) // ncol
    grid = torch.zeros(c, nr * h, ncol * w, dtype=torch.uint8)
    for i in range(n):
        r, q = divmod(i, ncol)
        grid[:, r * h:(r + 1) * h, q * w:(q + 1) * w] = x[i]
    Image.fromarray(grid.permute(1, 2, 0).numpy()).save(path)


def build_model(n_classes, device):
    net = ContextUnet(in_ch=Cfg.IN_CH, n_feat=Cfg.N_FEAT, n_classes=n_classes, bottleneck_k=Cfg.BOTTLENECK_K)
    return DDPM(nn_model=net, betas=Cfg.BETAS, n_T=Cfg.N_T, device=device, drop_prob=Cfg.DROP_PROB)


def split_indices(labels, val_split, seed=42):
    """Per-class (stratified) train / validation split of sample indices, like the reference's split (new_scripy.py:622-657)."""
    g = torch.Generator().manual_seed(seed)
    tr, va = [], []
    for cls in sorted(set(labels)):
        idx = [i for i, l in enumerate(labels) if l == cls]
        perm = [idx[j] for j in torch.randperm(len(idx), generator=g).tolist()]
        n_val = max(1, int(len(perm) * val_split)) if len(perm) > 1 else 0
        va += perm[:n_val]
        tr += perm[n_val:]
    return tr, va


def train_model(n_classes=4, n_train=64, n_val=16, max_epochs=None, device="cuda:0", quiet=False, data_root=None):
    os.makedirs(Cfg.SAVE_DIR, exist_ok=True)
    os.makedirs(Cfg.SAMPLE_DIR, exist_ok=True)
    S = Cfg.IMG_SIZE
    to_mask = lambda am: am.to(device)
    if data_root:       # the reference's layout (images/<class>/..., annotations/*.xml); masks are rasterised on the device
        from diffusionmodel_amd.data import CrackDataset, attn_masks
        full = CrackDataset(data_root, S, return_boxes=True)
        n_classes = len(full.classes)
        tr_idx, va_idx = split_indices([s[2] for s in full.samples], Cfg.VAL_SPLIT)
        train_ds, val_ds = torch.utils.data.Subset(full, tr_idx), torch.utils.data.Subset(full, va_idx)
        to_mask = lambda boxes: attn_masks(boxes.tolist(), S, device)
        if not quiet:
            print(f"{data_root}: {len(full)} images, classes {full.classes}, {len(tr_idx)} train / {len(va_idx)} val")
    else:
        train_ds = SyntheticCrackDataset(n_train, S, n_classes, seed=0)
        val_ds = SyntheticCrackDataset(n_val, S, n_classes, seed=1)
    train_dl = torch.utils.data.DataLoader(train_ds, batch_size=Cfg.BATCH_SIZE, shuffle=True, drop_last=True,
                                           num_workers=Cfg.NUM_WORKERS if data_root else 0)
    val_dl = torch.utils.data.DataLoader(val_ds, batch_size=Cfg.BATCH_SIZE)
    ddpm = build_model(n_classes, device)
    optim = FusedAdamW(ddpm.parameters(), lr=Cfg.LR, weight_decay=Cfg.WD, max_grad_norm=1.0)
    sched = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(optim, T_0=10, T_mult=2, eta_min=3e-5)
    stopper = EarlyStop(save_path=os.path.join(Cfg.SAVE_DIR, "best_model.pt"))
    history = []
    n_epoch = Cfg.N_EPOCH if max_epochs is None else max_epochs
    for ep in range(n_epoch):
        t0 = time.time()
        ddpm.train()
        optim.zero_grad()
        losses = []
        for it, (x, c, am) in enumerate(train_dl):
            loss = ddpm(x.to(device), c.to(device), to_mask(am)) / Cfg.ACCUM_STEPS          # :785-786
            loss.backward()
            losses.append(loss.detach())
            if (it + 1) % Cfg.ACCUM_STEPS == 0:                                            # :795-803
                optim.step()
                optim.zero_grad()
        ddpm.eval()                                                                        # :818-835
        with torch.no_grad():
            vl = [ddpm(x.to(device), c.to(device), to_mask(am)) for x, c, am in val_dl]
        tr = float(torch.stack(losses).mean()) * Cfg.ACCUM_STEPS
        va = float(torch.stack(vl).mean())
        history.append({"epoch": ep, "train_loss": tr, "val_loss": va, "lr": optim.param_groups[0]["lr"], "time": time.time() - t0})
        if not quiet:
            print(f"epoch {ep}: train {tr:.4f} val {va:.4f} lr {optim.param_groups[0]['lr']:.2e} ({time.time() - t0:.1f}s)")
        if stopper(va, ddpm, ep):                                                          # :838-845
            break
        sched.step()                                                                       # per epoch, :848
        if (ep + 1) % 5 == 0:                                                              # periodic sampling, :851-893
            for w in Cfg.GUIDE_SCALES:
                xs = ddpm.sample(n_classes * 2, (3, S, S), device, guide_w=w)
                save_samples(xs, os.path.join(Cfg.SAMPLE_DIR, f"ep{ep}_w{w}.png"), n_classes)
        if (ep + 1) % Cfg.SAVE_FREQ == 0 and ep + 1 >= Cfg.MIN_SAVE_EP or ep + 1 == n_epoch:     # :896-901
            torch.save({"epoch": ep, "model_state_dict": ddpm.state_dict(), "optimizer_state_dict": {"step": optim._step},
                        "scheduler_state_dict": sched.state_dict(), "loss": tr, "metrics": history},
                       os.path.join(Cfg.SAVE_DIR, f"ckpt_ep{ep}.pt"))
    with open(os.path.join(Cfg.SAVE_DIR, "metrics.json"), "w") as f:
        json.dump(history, f)
    return ddpm, history


def gen_samples(ckpt_path, n_samples_per_class=Cfg.SAMPLES_PER_CLASS, guide_scales=None, n_classes=4, device="cuda:0",
                use_graph=True):
    guide_scales = Cfg.GUIDE_SCALES if guide_scales is None else guide_scales
    os.makedirs(Cfg.SAMPLE_DIR, exist_ok=True)
    ddpm = build_model(n_classes, device)
    ddpm.drop_prob = 0.0
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    sd = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt   # :975-990
    ddpm.load_state_dict(sd)
    ddpm.eval()
    S = Cfg.IMG_SIZE
    out = {}
    for w in guide_scales:
        xs = ddpm.sample(n_samples_per_class * n_classes, (3, S, S), device, guide_w=w, use_graph=use_graph)
        path = os.path.join(Cfg.SAMPLE_DIR, f"generated_w{w}.png")
        save_samples(xs, path, n_classes)
        out[w] = xs
        print(f"guide scale {w}: wrote {path}")
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description="DDPM crack generator (MI355X drop-in)")
    ap.add_argument("--mode", choices=["train", "generate"], default="train")
    ap.add_argument("--ckpt", "--checkpoint", dest="ckpt", default=None)
    ap.add_argument("--guide_scales", "--guidance_scales", dest="guide_scales", type=float, nargs="+", default=None)
    ap.add_argument("--samples", "--samples_per_class", dest="samples", type=int, default=Cfg.SAMPLES_PER_CLASS)
    ap.add_argument("--no_eval", action="store_true", help="accepted for compatibility (quality metrics are out of scope)")
    ap.add_argument("--no_memory_cleanup", action="store_true", help="README flag; no-op")
    # additions: run-size knobs so the drivers are usable on synthetic data
    ap.add_argument("--img_size", type=int, default=None)
    ap.add_argument("--n_feat", type=int, default=None)
    ap.add_argument("--n_T", type=int, default=None)
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--batch_size", type=int, default=None)
    ap.add_argument("--dtype", choices=["float32", "bfloat16"], default=None)
    ap.add_argument("--bottleneck_k", type=int, default=None)
    ap.add_argument("--data_root", default=None, help="dataset in the reference's layout (images/<class>/*.jpg + annotations/*.xml); "
                                                      "default: synthetic tensors")
    ap.add_argument("--convert_supervisely", nargs=2, metavar=("SPLIT_DIR", "DST_ROOT"), default=None,
                    help="lay a DatasetNinja / Supervisely split (img/ + ann/*.json, e.g. the bundled road-damage set) out as --data_root and exit")
    a = ap.parse_args(argv)
    if a.convert_supervisely:
        from diffusionmodel_amd.data import convert_supervisely
        print(f"wrote {convert_supervisely(*a.convert_supervisely)} images under {a.convert_supervisely[1]}")
        return
    for name, val in (("IMG_SIZE", a.img_size), ("N_FEAT", a.n_feat), ("N_T", a.n_T), ("BATCH_SIZE", a.batch_size),
                      ("DTYPE", a.dtype), ("BOTTLENECK_K", a.bottleneck_k)):
        if val is not None:
            setattr(Cfg, name, val)
    if a.mode == "train":
        train_model(max_epochs=a.epochs, data_root=a.data_root)
    else:
        if not a.ckpt:
            ap.error("--mode generate needs --ckpt/--checkpoint")
        gen_samples(a.ckpt, a.samples, a.guide_scales)


if __name__ == "__main__":
    main()
