"""Evaluation helpers of the reference's drivers on the device (new_scripy.py:1111-1290, `ImageMetrics`).

`calc_ssim` (global-statistics SSIM, :1188-1222) and `calc_psnr` (:1224-1250) are computed from one HIP
reduction per image pair (`dm_image_moments`: first and second moments + minima in double precision);
the reference's "rescale to [0,1] if min < 0" rule is applied analytically to the moments.  FID needs
torchvision's pretrained InceptionV3 (a remote weight fetch) and is out of scope offline: `calc_fid`
raises, `evaluate_batch` reports NaN for it exactly where the reference would have tried.
"""
import math

import numpy as np
import torch

from ._lib import DmError, call, ptr, require_device


def image_moments(img1, img2):
    """(N, 8) float64 on the host: sum a, sum b, sum a^2, sum b^2, sum ab, min a, min b, n for every pair."""
    require_device(img1, img2)
    a = img1.float().contiguous().reshape(img1.shape[0] if img1.dim() == 4 else 1, -1)
    b = img2.float().contiguous().reshape(a.shape[0], -1)
    if a.shape != b.shape:
        raise DmError(f"image pair shapes differ: {tuple(img1.shape)} vs {tuple(img2.shape)}")
    out = torch.empty((a.shape[0], 8), dtype=torch.float64, device=a.device)
    call("dm_image_moments", ptr(a), ptr(b), ptr(out), a.shape[0], a.shape[1])
    return out.cpu().numpy()


def _rescaled(m):
    """Moments of a' = (a+1)/2 if min a < 0 else a (and the same for b): returns E[a'], E[b'], E[a'^2], E[b'^2], E[a'b']."""
    sa, sb, saa, sbb, sab, mna, mnb, n = m
    ea, eb, eaa, ebb, eab = sa / n, sb / n, saa / n, sbb / n, sab / n
    ka, oa = (0.5, 0.5) if mna < 0 else (1.0, 0.0)
    kb, ob = (0.5, 0.5) if mnb < 0 else (1.0, 0.0)
    ea2, eb2 = ka * ea + oa, kb * eb + ob
    eaa2 = ka * ka * eaa + 2 * ka * oa * ea + oa * oa
    ebb2 = kb * kb * ebb + 2 * kb * ob * eb + ob * ob
    eab2 = ka * kb * eab + ka * ob * ea + kb * oa * eb + oa * ob
    return ea2, eb2, eaa2, ebb2, eab2


def _ssim(m):
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    mu1, mu2, e11, e22, e12 = _rescaled(m)
    v1, v2, s12 = e11 - mu1 * mu1, e22 - mu2 * mu2, e12 - mu1 * mu2
    return ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 ** 2 + mu2 ** 2 + c1) * (v1 + v2 + c2))


def _psnr(m):
    _, _, e11, e22, e12 = _rescaled(m)
    mse = e11 - 2 * e12 + e22
    if mse <= 0:
        return float("inf")
    return 20 * math.log10(1.0 / math.sqrt(mse))


class ImageMetrics:
    """Same method names as the reference class; images are device tensors [C,H,W] (or batches [N,C,H,W])."""

    def __init__(self, device="cuda:0"):
        self.device = device

    @staticmethod
    def calc_ssim(img1, img2):
        return float(_ssim(image_moments(img1, img2)[0]))

    @staticmethod
    def calc_psnr(img1, img2):
        return float(_psnr(image_moments(img1, img2)[0]))

    def calc_fid(self, real_images, gen_images):
        raise DmError("FID needs torchvision's pretrained InceptionV3 (remote weight fetch): out of scope offline")

    def evaluate_batch(self, real_images, gen_images):
        metrics = {}
        if len(real_images) >= 10 and len(gen_images) >= 10:
            metrics["fid"] = float("nan")                     # the reference catches the failure and stores NaN (:1268-1272)
        if len(real_images) == len(gen_images):
            ms = image_moments(real_images, gen_images)       # one launch for the whole batch
            metrics["ssim"] = float(np.mean([_ssim(m) for m in ms]))
            metrics["psnr"] = float(np.mean([_psnr(m) for m in ms]))
        return metrics
