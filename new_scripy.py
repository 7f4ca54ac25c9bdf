#!/usr/bin/env python3
"""Drop-in for the reference's `new_scripy.py --mode train|generate` on MI355X.

The model / DDPM classes come from `diffusionmodel_amd` (HIP kernels); this file restates the two
drivers around them — `train_model()` (new_scripy.py:659-943: gradient accumulation, clip 1.0, AdamW,
CosineAnnealingWarmRestarts(10, 2, 3e-5) stepped per epoch, early stopping, checkpoint dict format,
periodic sampling) and `gen_samples()` (:945-1108: checkpoint load with raw-state-dict fallback,
class-cycled CFG sampling per guide scale, PNG grids) — and the CLI with BOTH flag spellings
(code: --ckpt --guide_scales --samples --no_eval; README: --checkpoint --guidance_scales
--samples_per_class --no_memory_cleanup).  File formats follow the reference: ckpt_ep{N}.pt / best_model.pt are the 6-key
dict with the optimiser's torch-AdamW-schema state (:730-744, :896-901), best_model_early.pt the 3-key early-stop dict
(:838-845), metrics/metrics_ep{N}.json the running log (:904-919); generation writes samples_g{w}.png, one PNG per image and
quality_metrics.json (SSIM / PSNR; FID needs torchvision's remote Inception weights and is out of scope).  Without
`--data_root` a synthetic dataset with the reference's mask-value convention (0.5 / 1.0 / 3.0, :535-546) is used.
"""
import argparse
import json
import os
import time

import numpy as np
import torch

from diffusionmodel_amd import Cfg, ContextUnet, DDPM, FusedAdamW
# every public name of the reference module resolves here too (new_scripy.py:70-268, 358, 479, 1111): `from new_scripy import ResConvBlock`
from diffusionmodel_amd import CoordAttn, EmbedFC, LocalEnhancer, ResConvBlock, SEBlock, UnetDown, UnetUp, ddpm_schedules  # noqa: F401
from diffusionmodel_amd import parallel
from diffusionmodel_amd.data import CrackDataset  # noqa: F401
from diffusionmodel_amd.metrics import ImageMetrics  # noqa: F401
from diffusionmodel_amd.train import StridedBatchSampler, TrainEngine, reduce_mean_scalar


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
    """Patience / min-delta convergence heuristic (new_scripy.py:587-620): returns True when the validation loss improved
    (the caller then writes best_model.pt), keeps the best state as {'epoch','model_state_dict','val_loss'} and raises
    `early_stop` after `patience` epochs without improvement."""

    def __init__(self, patience=Cfg.PATIENCE, min_delta=Cfg.MIN_DELTA, verbose=True):
        self.patience, self.min_delta, self.verbose = patience, min_delta, verbose
        self.counter, self.best_loss, self.early_stop, self.best_state = 0, float("inf"), False, None

    def __call__(self, val_loss, model, epoch):
        if val_loss < self.best_loss - self.min_delta:
            self.best_loss, self.counter = val_loss, 0
            if self.verbose:
                print(f"Val loss improved to {val_loss:.6f}")
            sd = model.state_dict() if model is not None else None
            if sd is not None:                     # parameters are views into the optimiser's flat buffer: keep a snapshot, not the views
                sd = {k: v.detach().clone() for k, v in sd.items()}
            self.best_state = {"epoch": epoch, "model_state_dict": sd, "val_loss": val_loss}
            return True
        self.counter += 1
        if self.verbose:
            print(f"Val loss not improved, patience: {self.counter}/{self.patience}")
        if self.counter >= self.patience:
            self.early_stop = True
            if self.verbose:
                print("Early stopping triggered! Training halted.")
        return False


def _to_uint8(x, denorm=True):
    x = x.detach().float().cpu()
    x = x * 0.5 + 0.5 if denorm else x                                  # new_scripy.py:556-557
    return (x.clamp(0, 1) * 255 + 0.5).to(torch.uint8)                  # torchvision.utils.save_image's quantisation


def save_image(img, path, denorm=True):
    """One (C,H,W) image as a PNG (the reference's per-image save_image, new_scripy.py:1044-1061)."""
    from PIL import Image
    a = _to_uint8(img, denorm).permute(1, 2, 0).numpy()
    Image.fromarray(a[:, :, 0] if a.shape[2] == 1 else a).save(path)


def save_samples(x, path, nrow=None, denorm=True, padding=2):
    """PNG grid without torchvision (new_scripy.py:554-561 used make_grid/save_image): `nrow` images per row (make_grid's
    meaning of nrow; default 8), 2-pixel black padding like make_grid's default."""
    from PIL import Image
    x = _to_uint8(x, denorm)
    n, c, h, w = x.shape
    ncol = min(nrow or 8, n)
    nr = (n + ncol - 1) // ncol
    grid = torch.zeros(c, nr * (h + padding) + padding, ncol * (w + padding) + padding, dtype=torch.uint8)
    for i in range(n):
        r, q = divmod(i, ncol)
        y0, x0 = padding + r * (h + padding), padding + q * (w + padding)
        grid[:, y0:y0 + h, x0:x0 + w] = x[i]
    a = grid.permute(1, 2, 0).numpy()
    Image.fromarray(a[:, :, 0] if c == 1 else a).save(path)
    return path


def build_model(n_classes, device, drop_prob=None):
    net = ContextUnet(in_ch=Cfg.IN_CH, n_feat=Cfg.N_FEAT, n_classes=n_classes, bottleneck_k=Cfg.BOTTLENECK_K)
    return DDPM(nn_model=net, betas=Cfg.BETAS, n_T=Cfg.N_T, device=device, drop_prob=Cfg.DROP_PROB if drop_prob is None else drop_prob)


def split_indices(labels, val_split, seed=42):
    """Stratified train / validation split of sample indices (new_scripy.py:628-633).  With scikit-learn importable — the
    reference's own dependency — it IS the reference's split: StratifiedShuffleSplit(n_splits=1, test_size=val_split,
    random_state=42); otherwise (or when a class has a single member, which sklearn refuses) a per-class seeded permutation."""
    try:
        from sklearn.model_selection import StratifiedShuffleSplit
        tr, va = next(StratifiedShuffleSplit(n_splits=1, test_size=val_split, random_state=seed).split(np.zeros(len(labels)), labels))
        return [int(i) for i in tr], [int(i) for i in va]
    except (ImportError, ValueError):
        pass
    g = torch.Generator().manual_seed(seed)
    tr, va = [], []
    for cls in sorted(set(labels)):
        idx = [i for i, l in enumerate(labels) if l == cls]
        perm = [idx[j] for j in torch.randperm(len(idx), generator=g).tolist()]
        n_val = max(1, int(len(perm) * val_split)) if len(perm) > 1 else 0
        va += perm[:n_val]
        tr += perm[n_val:]
    return tr, va


def create_loaders(dataset, batch_size, val_split=Cfg.VAL_SPLIT, num_workers=Cfg.NUM_WORKERS, pin_mem=Cfg.PIN_MEM, *, rank=0, world=1, seed=0):
    """new_scripy.py:622-657: stratified split of `dataset` (labels = dataset.samples[i][2]) and the two loaders — train shuffled,
    validation in order, no drop_last.  rank / world (keyword-only additions): the loaders of ONE rank of a data-parallel job —
    micro-batch m of the epoch's single seeded permutation goes to rank m % world (train.StridedBatchSampler; call
    `train_loader.batch_sampler.set_epoch(ep)` per epoch) — with world == 1 they are the reference's loaders."""
    from torch.utils.data import DataLoader, Subset
    labels = [dataset.samples[i][2] for i in range(len(dataset))]
    train_idx, val_idx = split_indices(labels, val_split)
    train_set, val_set = Subset(dataset, train_idx), Subset(dataset, val_idx)
    print(f"Dataset split - Train: {len(train_set)} samples, Val: {len(val_set)} samples")
    if world > 1:
        tb = StridedBatchSampler(len(train_set), batch_size, rank, world, shuffle=True, seed=seed)
        vb = StridedBatchSampler(len(val_set), batch_size, rank, world, shuffle=False)
        return (DataLoader(train_set, batch_sampler=tb, num_workers=num_workers, pin_memory=pin_mem),
                DataLoader(val_set, batch_sampler=vb, num_workers=num_workers, pin_memory=pin_mem))
    return (DataLoader(train_set, batch_size=batch_size, shuffle=True, num_workers=num_workers, pin_memory=pin_mem),
            DataLoader(val_set, batch_size=batch_size, shuffle=False, num_workers=num_workers, pin_memory=pin_mem))


def gen_and_save(model, n_samples, img_size, device, guide_scales, save_dir, classes=None, denorm=True, samples_per_class=None):
    """new_scripy.py:563-585: one `model.sample` call per guide scale (samples_per_class x len(classes) images when
    samples_per_class is given, else n_samples), the grid written to save_dir/samples_g{w}.png with samples_per_class images per
    row; returns {w: {"samples", "grid_path"}}."""
    results = {}
    n_classes = len(classes) if classes else 1
    with torch.no_grad():
        for guide_scale in guide_scales:
            print(f"\nGenerating samples with guidance scale {guide_scale}")
            n_sample = samples_per_class * n_classes if samples_per_class else n_samples
            x_gen = model.sample(n_sample, (Cfg.IN_CH, *img_size), device, guide_w=guide_scale)
            grid_path = os.path.join(save_dir, f"samples_g{guide_scale}.png")
            save_samples(x_gen, grid_path, nrow=samples_per_class, denorm=denorm)
            results[guide_scale] = {"samples": x_gen, "grid_path": grid_path}
    return results


def _jsonable(v):
    if isinstance(v, dict):
        return {str(k): _jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, (np.floating, np.integer)):
        return float(v)
    return v


def _dist_setup(device):
    """(rank, world, device): joins the process group torchrun / --gpus N described (RANK, WORLD_SIZE, MASTER_*), one process per
    GPU, `nccl` = RCCL (DM_DIST_BACKEND=gloo: rehearsal on fewer devices, ranks then share them).  Without WORLD_SIZE: (0, 1, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        Cfg.WORLD_SIZE = 1
        return 0, 1, device
    backend = os.environ.get("DM_DIST_BACKEND") or None
    if backend == "gloo" and torch.cuda.is_available():
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
    rank, world, local = parallel.init_from_env(backend)
    torch.cuda.set_device(local)
    Cfg.WORLD_SIZE = world
    return rank, world, f"cuda:{local}"


def _sample_all_ranks(ddpm, n, size, device, w, seed, world, **kw):
    """ddpm.sample over the ranks of the job: shards of n / world samples with their slice of the noise stream and one gather
    (parallel.sample_sharded) when n divides; otherwise every rank computes all n from the same seed (identical images)."""
    if world > 1 and n % world == 0:
        return parallel.sample_sharded(ddpm, n, size, device, guide_w=w, seed=seed, **kw)
    return ddpm.sample(n_sample=n, size=size, device=device, guide_w=w, seed=seed if world > 1 else None, **kw)


def train_model(n_classes=4, n_train=64, n_val=16, max_epochs=None, device="cuda:0", quiet=False, data_root=None, use_plan=True):
    """new_scripy.py:659-943.  Returns (ddpm, per-epoch history).

    Under torchrun / `--gpus N` (WORLD_SIZE > 1) the loop is data parallel, one process per GPU: micro-batch m of the epoch goes to
    rank m % world, an optimiser step covers world * ceil(ACCUM_STEPS / world) micro-batches weighed like the reference's
    accumulation group (:786, :795-803; diffusionmodel_amd/train.py), the flat gradient is all-reduced over RCCL in buckets that
    overlap the backward pass, validation batches are sharded the same way, rank 0's BatchNorm statistics are the model's
    (broadcast before every validation / sampling pass and the ones checkpointed), periodic sampling is sharded over the ranks
    and only rank 0 writes files.  A step that is one fixed-shape micro-batch per rank runs as a launch plan."""
    from diffusionmodel_amd.metrics import ImageMetrics
    rank, world, device = _dist_setup(device)
    chief = rank == 0
    say = (lambda *a: None) if (quiet or not chief) else print
    metrics_dir = os.path.join(Cfg.SAVE_DIR, "metrics")                                    # :664-665
    if chief:
        os.makedirs(Cfg.SAVE_DIR, exist_ok=True)
        os.makedirs(metrics_dir, exist_ok=True)
    metrics_log = {"train_loss": [], "val_loss": [], "img_metrics": [], "lr": []}          # :668-673
    guide_scales = Cfg.GUIDE_SCALES
    img_metrics = ImageMetrics(device=device)
    S = Cfg.IMG_SIZE
    to_mask = lambda am: am.to(device)
    if data_root:       # the reference's layout (images/<class>/..., annotations/*.xml); masks are rasterised on the device
        from diffusionmodel_amd.data import attn_masks
        full = CrackDataset(data_root, S, return_boxes=True)
        n_classes = len(full.classes)
        tr_idx, va_idx = split_indices([s[2] for s in full.samples], Cfg.VAL_SPLIT)
        train_ds, val_ds = torch.utils.data.Subset(full, tr_idx), torch.utils.data.Subset(full, va_idx)
        to_mask = lambda boxes: attn_masks(boxes.tolist(), S, device)
        say(f"{data_root}: {len(full)} images, classes {full.classes}, {len(tr_idx)} train / {len(va_idx)} val")
    else:
        train_ds = SyntheticCrackDataset(n_train, S, n_classes, seed=0)
        val_ds = SyntheticCrackDataset(n_val, S, n_classes, seed=1)
    # no drop_last: the reference flushes a short tail group instead (:795).  One seeded permutation per epoch, shared by the ranks
    seed = [int(torch.initial_seed()) & 0x7FFFFFFF]
    if world > 1:
        torch.distributed.broadcast_object_list(seed, src=0)
    train_bs = StridedBatchSampler(len(train_ds), Cfg.BATCH_SIZE, rank, world, shuffle=True, seed=seed[0])
    val_bs = StridedBatchSampler(len(val_ds), Cfg.BATCH_SIZE, rank, world, shuffle=False)
    train_dl = torch.utils.data.DataLoader(train_ds, batch_sampler=train_bs, num_workers=Cfg.NUM_WORKERS if data_root else 0)
    val_dl = torch.utils.data.DataLoader(val_ds, batch_sampler=val_bs)
    torch.manual_seed(seed[0])                                                             # identical initial weights on every rank
    ddpm = build_model(n_classes, device)
    ddpm.rng_seed = seed[0] + 7919 * rank                                                  # every rank draws its own t / noise / keep masks
    shadow = ddpm.compute_dtype if ddpm.compute_dtype != torch.float32 else torch.bfloat16
    optim = FusedAdamW(ddpm.parameters(), lr=Cfg.LR, weight_decay=Cfg.WD, max_grad_norm=1.0, shadow_dtype=shadow)   # :715-719 + clip of :798
    if world > 1:
        from diffusionmodel_amd import ops as _ops
        parallel.broadcast_parameters(optim.flat_p)
        optim.refresh_shadow()
        _ops.bump_weight_epoch()
        parallel.broadcast_buffers(ddpm)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(optim, T_0=10, T_mult=2, eta_min=3e-5)
    engine = TrainEngine(ddpm, optim, accum_steps=Cfg.ACCUM_STEPS, use_plan=use_plan)
    if world > 1 and engine.G != Cfg.ACCUM_STEPS:
        say(f"[data parallel] {world} ranks, ACCUM_STEPS={Cfg.ACCUM_STEPS}: an optimiser step covers {engine.G} micro-batches "
            f"({engine.local_accum} per rank)")
    if train_bs.n_batches < world:
        raise SystemExit(f"{train_bs.n_batches} micro-batches per epoch cannot feed {world} ranks")
    early_stop = EarlyStop(patience=Cfg.PATIENCE, min_delta=Cfg.MIN_DELTA, verbose=not quiet and chief)
    n_epoch = Cfg.N_EPOCH if max_epochs is None else max_epochs

    def save_ckpt(epoch, loss, is_best=False):                                             # :730-744
        if not chief:
            return
        path = os.path.join(Cfg.SAVE_DIR, "best_model.pt" if is_best else f"ckpt_ep{epoch}.pt")
        torch.save({"epoch": epoch, "model_state_dict": ddpm.state_dict(), "optimizer_state_dict": optim.state_dict(),
                    "scheduler_state_dict": scheduler.state_dict(), "loss": loss, "metrics": _jsonable(metrics_log)}, path)
        say(f'Saved {"best " if is_best else ""}checkpoint: {path}')

    # validation samples for the periodic quality evaluation (:746-765); every rank collects the same ones (it shards the sampling)
    eval_samples, eval_count = [], min(32, len(val_ds))
    per_class = max(2, eval_count // n_classes)
    counts = {i: 0 for i in range(n_classes)}
    for i in range(len(val_ds)):
        x_i, c_i = val_ds[i][0], int(val_ds[i][1])
        if counts[c_i] < per_class and len(eval_samples) < eval_count:
            eval_samples.append((x_i.to(device), c_i))
            counts[c_i] += 1
        if sum(counts.values()) >= eval_count:
            break
    say(f"Collected {len(eval_samples)} samples for evaluation")

    history, train_loss_ema = [], None
    for ep in range(n_epoch):
        t0 = time.time()
        ddpm.train()
        train_bs.set_epoch(ep)
        train_loss_ema, losses = None, []
        it = iter(train_dl)
        for s in range(train_bs.slots):
            if train_bs.mine[s] is None:                                                   # short tail: the peers have a micro-batch here
                engine.idle_slot()
                continue
            x, c, am = next(it)
            # :784-803 — loss / ACCUM, scaled backward, and at the end of a group (or of the epoch, :795) unscale + clip + AdamW
            losses.append(engine.micro_batch(x.to(device), c.long().to(device), to_mask(am), last_in_epoch=s == train_bs.slots - 1))
        vals = [float(v) * engine.loss_div for v in torch.stack(losses).reshape(-1).tolist()]   # one readback per epoch (:789 reads every step)
        if world > 1:                                 # the epoch's micro-batch losses in micro-batch order, on every rank
            allv = [None] * world
            torch.distributed.all_gather_object(allv, vals)
            vals = [allv[m % world][m // world] for m in range(train_bs.n_batches)]
        for v in vals:
            train_loss_ema = v if train_loss_ema is None else 0.95 * train_loss_ema + 0.05 * v          # :806-809
        tr = sum(vals) / len(vals)
        metrics_log["train_loss"].append(tr)
        metrics_log["lr"].append(scheduler.get_last_lr()[0])
        parallel.broadcast_buffers(ddpm)              # rank 0's BatchNorm statistics are the model's
        ddpm.eval()                                                                        # :818-835
        with torch.no_grad():
            vl = [ddpm(x.to(device), c.long().to(device), to_mask(am)) for x, c, am in val_dl]
        va = reduce_mean_scalar(float(torch.stack(vl).sum()) if vl else 0.0, len(vl), device)
        metrics_log["val_loss"].append(va)
        history.append({"epoch": ep, "train_loss": tr, "val_loss": va, "lr": optim.param_groups[0]["lr"], "time": time.time() - t0})
        say(f"epoch {ep}: train {tr:.4f} val {va:.4f} lr {optim.param_groups[0]['lr']:.2e} ({time.time() - t0:.1f}s)")
        is_best = early_stop(va, ddpm if chief else None, ep)                              # :838
        if world > 1:
            # every rank saw the same all-reduced `va`, so the decisions agree; rank 0's is nevertheless the one that counts — a
            # rank that stopped alone would leave its peers waiting in the next step's collectives
            flags = [bool(is_best), bool(early_stop.early_stop)]
            torch.distributed.broadcast_object_list(flags, src=0)
            is_best, early_stop.early_stop = flags
        if early_stop.early_stop:                                                          # :839-845
            if chief and early_stop.best_state:
                torch.save(early_stop.best_state, os.path.join(Cfg.SAVE_DIR, "best_model_early.pt"))
            break
        scheduler.step()                                                                   # per epoch, :848
        if (ep % 5 == 0 or ep == n_epoch - 1) and eval_samples:                            # :851-893
            ddpm.eval()
            n_gen = len(eval_samples) // n_classes * n_classes      # sample() needs a multiple of n_classes (:448)
            real = torch.stack([s[0] for s in eval_samples])[:n_gen]
            for gi, w in enumerate(guide_scales):
                if n_gen == 0:
                    break
                x_gen = _sample_all_ranks(ddpm, n_gen, (Cfg.IN_CH, S, S), device, w, seed[0] + 1000 * ep + gi + 1, world)
                if not chief:
                    continue
                grid_path = os.path.join(Cfg.SAVE_DIR, f"img_ep{ep}_w{w}.png")
                save_samples(x_gen, grid_path, nrow=4)
                try:
                    q = img_metrics.evaluate_batch(real, x_gen)
                    q["guide_scale"], q["epoch"] = w, ep
                    metrics_log["img_metrics"].append(q)
                    say(f"Image quality metrics (w={w}): " + ", ".join(f"{k.upper()} {v:.4f}" for k, v in q.items() if k not in ("guide_scale", "epoch")))
                except Exception as e:                                                     # noqa: BLE001 — as the reference (:892)
                    say(f"Quality assessment failed: {e}")
        if ((ep + 1) % Cfg.SAVE_FREQ == 0 or ep == n_epoch - 1) and ep >= Cfg.MIN_SAVE_EP:  # :896-897
            save_ckpt(ep, train_loss_ema)
        if is_best:                                                                        # :900-901
            save_ckpt(ep, va, is_best=True)
        if chief:
            with open(os.path.join(metrics_dir, f"metrics_ep{ep}.json"), "w") as f:        # :904-919
                json.dump(_jsonable(metrics_log), f, indent=2)
    save_ckpt(n_epoch - 1, train_loss_ema)                                                 # final model, :931
    if world > 1:                                     # every rank leaves with the best weights rank 0 kept (:934-936)
        have = [bool(early_stop.best_state and early_stop.best_state["model_state_dict"] is not None)]
        torch.distributed.broadcast_object_list(have, src=0)
        if have[0]:
            if chief:
                ddpm.load_state_dict(early_stop.best_state["model_state_dict"])
            parallel.broadcast_parameters(optim.flat_p)
            parallel.broadcast_buffers(ddpm)
            optim.refresh_shadow()
    elif early_stop.best_state and early_stop.best_state["model_state_dict"] is not None:  # :934-936
        ddpm.load_state_dict(early_stop.best_state["model_state_dict"])
        optim.refresh_shadow()
    if world > 1:
        # the ranks must leave with the same model: one tiny collective (sum and sum of squares of the flat parameter buffer in
        # float64) — a mismatch is a bug in the reduction / broadcast path and must not pass silently
        chk = torch.stack([optim.flat_p.double().sum(), (optim.flat_p.double() ** 2).sum()]).cpu()
        allc = [None] * world
        torch.distributed.all_gather_object(allc, [float(v) for v in chk])
        if any(a != allc[0] for a in allc):
            raise RuntimeError(f"[data parallel] parameters differ between ranks at exit: {allc}")
        say(f"[data parallel] parameters identical on {world} ranks (checksum {allc[0][0]:.9e} / {allc[0][1]:.9e})")
    ddpm.train_engine = engine
    return ddpm, history


def _classes_of_checkpoint(sd):
    """The class count a checkpoint was trained with: input width of the context embedding (the reference rebuilds the dataset
    just to count its classes, new_scripy.py:960-961)."""
    for k in ("nn_model.ctx_emb1.model.0.weight", "ctx_emb1.model.0.weight"):
        if k in sd:
            return int(sd[k].shape[1])
    raise KeyError("checkpoint has no ctx_emb1.model.0.weight: not a ContextUnet / DDPM state dict")


def gen_samples(ckpt_path, n_samples_per_class=Cfg.SAMPLES_PER_CLASS, guide_scales=None, denorm=True, eval_quality=True,
                n_classes=None, class_names=None, real_images=None, device="cuda:0", use_graph=True, data_root=None, seed=None, save_raw=False):
    """new_scripy.py:945-1108.  Class count / names come from `data_root` (the reference's dataset layout) when given, otherwise
    from the checkpoint itself; per-image PNGs are named <class>_s<k>_g<w>.png; SSIM / PSNR against `real_images` (or the first
    images of `data_root`) unless eval_quality is off."""
    guide_scales = Cfg.GUIDE_SCALES if guide_scales is None else guide_scales
    S = Cfg.IMG_SIZE
    rank, world, device = _dist_setup(device)         # torchrun / --gpus N: the samples of every guide scale are sharded over the ranks
    chief = rank == 0
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    sd = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt   # :975-990
    dataset = None
    if data_root:
        from diffusionmodel_amd.data import CrackDataset
        dataset = CrackDataset(data_root, S)
        class_names = list(dataset.classes)
    if n_classes is None:
        n_classes = len(class_names) if class_names else _classes_of_checkpoint(sd)
    class_names = list(class_names) if class_names else [f"class{i}" for i in range(n_classes)]
    ddpm = build_model(n_classes, device, drop_prob=0.0)
    ddpm.load_state_dict(sd)
    if chief and isinstance(ckpt, dict) and "metrics" in ckpt:
        print("Checkpoint contains training metrics")
    ddpm.eval()
    stamp = [int(time.time())]
    if world > 1:
        torch.distributed.broadcast_object_list(stamp, src=0)
    samples_dir = os.path.join(Cfg.SAMPLE_DIR, f"samples_{stamp[0]}")                      # :996-998
    if chief:
        os.makedirs(samples_dir, exist_ok=True)
    if eval_quality and real_images is None and dataset is not None:                       # :1001-1029
        need = n_samples_per_class * min(n_classes, 4)
        real_images = torch.stack([dataset[i][0] for i in range(min(need, len(dataset)))])
    if real_images is not None:
        real_images = real_images.to(device)
    from diffusionmodel_amd.metrics import ImageMetrics
    img_metrics = ImageMetrics(device=device)
    results, quality = {}, {}
    for gi, w in enumerate(guide_scales):
        n_sample = n_samples_per_class * n_classes
        # `seed` (addition; the reference never seeds): the images are then a function of (checkpoint, seed, guide scale) alone —
        # whatever the number of ranks (the shards draw their slice of one noise stream)
        if seed is None:
            x_gen = _sample_all_ranks(ddpm, n_sample, (Cfg.IN_CH, S, S), device, w, stamp[0] + gi, world, use_graph=use_graph)   # :1036-1041
        else:
            x_gen = parallel.sample_sharded(ddpm, n_sample, (Cfg.IN_CH, S, S), device, guide_w=w, seed=int(seed) + gi, use_graph=use_graph)
        if chief and save_raw:
            torch.save(x_gen.detach().float().cpu(), os.path.join(samples_dir, f"samples_g{w}.pt"))
        grid_path = os.path.join(samples_dir, f"samples_g{w}.png")
        results[w] = {"samples": x_gen, "grid_path": grid_path}
        if not chief:
            continue
        save_samples(x_gen, grid_path, nrow=n_samples_per_class, denorm=denorm)             # :1040-1043
        for i in range(len(x_gen)):                                                        # :1045-1055 (file naming as the reference)
            name = class_names[(i // n_samples_per_class) % n_classes]
            save_image(x_gen[i], os.path.join(samples_dir, f"{name}_s{i % n_samples_per_class}_g{w}.png"), denorm)
        if eval_quality and real_images is not None and len(real_images) > 0:              # :1058-1068
            k = min(len(real_images), len(x_gen))
            quality[w] = img_metrics.evaluate_batch(real_images[:k], x_gen[:k])
            print(f"Image quality metrics (w={w}): " + ", ".join(f"{m.upper()} {v:.4f}" for m, v in quality[w].items()))
        results[w] = {"samples": x_gen, "grid_path": grid_path}
        print(f"guide scale {w}: wrote {grid_path}")
    if chief and eval_quality and quality:                                                 # :1075-1093
        with open(os.path.join(samples_dir, "quality_metrics.json"), "w") as f:
            json.dump(_jsonable(quality), f, indent=2)
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(description="DDPM crack generator (MI355X drop-in)")
    ap.add_argument("--mode", choices=["train", "generate"], default="train")
    ap.add_argument("--ckpt", "--checkpoint", dest="ckpt", default=None)
    ap.add_argument("--guide_scales", "--guidance_scales", dest="guide_scales", type=float, nargs="+", default=None)
    ap.add_argument("--samples", "--samples_per_class", dest="samples", type=int, default=Cfg.SAMPLES_PER_CLASS)
    ap.add_argument("--no_eval", action="store_true", help="skip the SSIM / PSNR pass of --mode generate")
    ap.add_argument("--no_memory_cleanup", action="store_true", help="README flag; no-op")
    # additions: run-size knobs so the drivers are usable on synthetic data
    ap.add_argument("--img_size", type=int, default=None)
    ap.add_argument("--n_feat", type=int, default=None)
    ap.add_argument("--n_T", type=int, default=None)
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--batch_size", type=int, default=None)
    ap.add_argument("--dtype", choices=["float32", "bfloat16", "float16"], default=None)
    ap.add_argument("--bottleneck_k", type=int, default=None)
    ap.add_argument("--accum_steps", type=int, default=None)
    ap.add_argument("--patience", type=int, default=None)
    ap.add_argument("--min_delta", type=float, default=None)
    ap.add_argument("--save_dir", default=None)
    ap.add_argument("--sample_dir", default=None)
    ap.add_argument("--n_train", type=int, default=64, help="size of the synthetic training set (no --data_root)")
    ap.add_argument("--n_val", type=int, default=16, help="size of the synthetic validation set (no --data_root)")
    ap.add_argument("--seed", type=int, default=None, help="--mode generate: seed of the sampling noise (default: the wall clock, like the unseeded reference)")
    ap.add_argument("--save_raw", action="store_true", help="--mode generate: also write the float32 images of every guide scale as samples_g{w}.pt")
    ap.add_argument("--gpus", type=int, default=None, help="start this many ranks (one process per GPU, RCCL data parallel); the same as "
                                                           "running the script under torch.distributed.run --nproc-per-node N")
    ap.add_argument("--no_plan", action="store_true", help="issue every training step eagerly (default: a fixed-shape step is captured once "
                                                           "and replayed as a launch plan)")
    ap.add_argument("--data_root", default=None, help="dataset in the reference's layout (images/<class>/*.jpg + annotations/*.xml); "
                                                      "default: synthetic tensors")
    ap.add_argument("--convert_supervisely", nargs=2, metavar=("SPLIT_DIR", "DST_ROOT"), default=None,
                    help="lay a DatasetNinja / Supervisely split (img/ + ann/*.json, e.g. the bundled road-damage set) out as --data_root and exit")
    a = ap.parse_args(argv)
    if a.gpus and a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # be the launcher: this process has made no GPU call; the children are this script, one rank per GPU
        import sys
        rest = list(sys.argv[1:] if argv is None else argv)
        i = rest.index("--gpus")
        del rest[i:i + 2]
        raise SystemExit(parallel.launch_ranks(__file__, rest, a.gpus))
    if a.convert_supervisely:
        from diffusionmodel_amd.data import convert_supervisely
        print(f"wrote {convert_supervisely(*a.convert_supervisely)} images under {a.convert_supervisely[1]}")
        return
    for name, val in (("IMG_SIZE", a.img_size), ("N_FEAT", a.n_feat), ("N_T", a.n_T), ("BATCH_SIZE", a.batch_size),
                      ("DTYPE", a.dtype), ("BOTTLENECK_K", a.bottleneck_k), ("ACCUM_STEPS", a.accum_steps), ("PATIENCE", a.patience),
                      ("MIN_DELTA", a.min_delta), ("SAVE_DIR", a.save_dir), ("SAMPLE_DIR", a.sample_dir)):
        if val is not None:
            setattr(Cfg, name, val)
    if a.mode == "train":
        kw = {"use_plan": False} if a.no_plan else {}
        if (a.n_train, a.n_val) != (64, 16):
            kw.update(n_train=a.n_train, n_val=a.n_val)
        if a.guide_scales:
            Cfg.GUIDE_SCALES = list(a.guide_scales)
        train_model(max_epochs=a.epochs, data_root=a.data_root, **kw)
    else:
        if not a.ckpt:
            ap.error("--mode generate needs --ckpt/--checkpoint")
        kw = {}
        if a.seed is not None or a.save_raw:
            kw = {"seed": a.seed, "save_raw": a.save_raw}
        gen_samples(a.ckpt, n_samples_per_class=a.samples, guide_scales=a.guide_scales, eval_quality=not a.no_eval, data_root=a.data_root, **kw)


if __name__ == "__main__":
    main()
