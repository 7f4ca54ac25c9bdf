#!/usr/bin/env python3
"""Drop-in for the reference's `MNIST_script.py` (BASELINE configs[0]: 28x28, 10 classes, the tutorial the project grew out of)
on MI355X: `train_mnist()` restated (MNIST_script.py:303-394) around `diffusionmodel_amd.mnist.{ContextUnet, DDPM}`.

What stays: the hard-coded hyper-parameters as defaults (20 epochs, batch 256, n_T = 400, n_feat = 128, lr 1e-4 with the linear decay
per epoch, Adam without weight decay or clipping, drop_prob 0.1), the loss EMA, the per-epoch evaluation (4 samples per class at
w in {0, 0.5, 2}, generated rows over real rows, inverted grey grid `image_ep{ep}_w{w}.png`), the optional `model_{ep}.pth`.
What differs: `MNIST("./data", download=True)` (:325) is a network fetch — without `data=(images, labels)` a synthetic set of
class-dependent blobs in [0, 1] stands in; the matplotlib GIF animation (:369-388) is plotting and out of scope.
"""
import os

import torch

from diffusionmodel_amd import FusedAdamW
from diffusionmodel_amd.mnist import DDPM, ContextUnet
# the other public names of the reference module (MNIST_script.py:31-216): `from MNIST_script import ResidualConvBlock` works
from diffusionmodel_amd.mnist import ResidualConvBlock, UnetDown, UnetUp  # noqa: F401
from diffusionmodel_amd import EmbedFC, ddpm_schedules  # noqa: F401


def synthetic_digits(n, n_classes=10, seed=0):
    """(n,1,28,28) images in [0,1] whose structure depends on the label (a blob whose position / size follows the class)."""
    g = torch.Generator().manual_seed(seed)
    labels = torch.randint(0, n_classes, (n,), generator=g)
    yy, xx = torch.meshgrid(torch.arange(28.0), torch.arange(28.0), indexing="ij")
    imgs = torch.empty(n, 1, 28, 28)
    for i in range(n):
        k = int(labels[i])
        cy, cx = 8 + 12 * ((k // 5) % 2) + float(torch.randn((), generator=g)), 4 + 5 * (k % 5) + float(torch.randn((), generator=g))
        sig = 2.0 + 0.3 * k
        imgs[i, 0] = torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sig * sig))
    return imgs.clamp_(0, 1), labels


def _save_grid(x_all, path, nrow=10):
    from PIL import Image
    x = (x_all.detach().float().cpu() * -1 + 1).clamp(0, 1)            # MNIST_script.py:365: make_grid(x_all*-1 + 1, nrow=10)
    n, _, h, w = x.shape
    rows = (n + nrow - 1) // nrow
    grid = torch.zeros(rows * (h + 2) + 2, nrow * (w + 2) + 2)
    for i in range(n):
        r, q = divmod(i, nrow)
        grid[2 + r * (h + 2):2 + r * (h + 2) + h, 2 + q * (w + 2):2 + q * (w + 2) + w] = x[i, 0]
    Image.fromarray((grid * 255 + 0.5).to(torch.uint8).numpy()).save(path)


def train_mnist(n_epoch=20, batch_size=256, n_T=400, device="cuda:0", n_classes=10, n_feat=128, lrate=1e-4, save_model=False,
                save_dir="./data/diffusion_outputs10/", ws_test=(0.0, 0.5, 2.0), data=None, n_synth=2048, dtype=None, quiet=False):
    os.makedirs(save_dir, exist_ok=True)
    ddpm = DDPM(nn_model=ContextUnet(in_channels=1, n_feat=n_feat, n_classes=n_classes, dtype=dtype), betas=(1e-4, 0.02), n_T=n_T,
                device=device, drop_prob=0.1)
    images, labels = data if data is not None else synthetic_digits(n_synth, n_classes)
    ds = torch.utils.data.TensorDataset(images, labels)
    loader = torch.utils.data.DataLoader(ds, batch_size=batch_size, shuffle=True)
    # torch.optim.Adam(lr) of the reference (:327) = AdamW without decay; no clipping in this script
    shadow = ddpm.compute_dtype if ddpm.compute_dtype != torch.float32 else torch.bfloat16
    optim = FusedAdamW(ddpm.parameters(), lr=lrate, weight_decay=0.0, max_grad_norm=None, shadow_dtype=shadow)
    history = []
    for ep in range(n_epoch):
        ddpm.train()
        optim.param_groups[0]["lr"] = lrate * (1 - ep / n_epoch)                         # linear lrate decay (:334)
        loss_ema, losses = None, []
        for x, c in loader:
            optim.zero_grad()
            x, c = x.to(device), c.to(device)
            loss = ddpm(x, c)
            ddpm.scaler.scale(loss).backward()
            ddpm.scaler.step(optim)
            losses.append(loss.detach())
        for v in torch.stack(losses).tolist():                                           # one readback per epoch (:342-345 reads every step)
            loss_ema = v if loss_ema is None else 0.95 * loss_ema + 0.05 * v
        history.append(loss_ema)
        if not quiet:
            print(f"epoch {ep}: loss {loss_ema:.4f}")
        ddpm.eval()
        with torch.no_grad():                                                            # :350-367
            n_sample = 4 * n_classes
            for w in ws_test:
                x_gen, x_gen_store = ddpm.sample(n_sample, (1, 28, 28), device, guide_w=w)
                x_real = torch.zeros_like(x_gen)
                for k in range(n_classes):
                    idx_k = (c == k).nonzero().flatten()
                    for j in range(n_sample // n_classes):
                        x_real[k + j * n_classes] = x[int(idx_k[j]) if j < len(idx_k) else 0]
                _save_grid(torch.cat([x_gen, x_real]), os.path.join(save_dir, f"image_ep{ep}_w{w}.png"))
        if save_model and ep == n_epoch - 1:
            torch.save(ddpm.state_dict(), os.path.join(save_dir, f"model_{ep}.pth"))
    return ddpm, history


if __name__ == "__main__":
    train_mnist()
