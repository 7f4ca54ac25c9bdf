"""Attention-mask generation of the reference's CrackDataset on the device (new_scripy.py:533-546).

The VOC-XML / image I/O of the dataset stays out of scope (host I/O); what the loss consumes —
the (B,S,S) mask with values LOW / MID (lower half) / HIGH (inside the scaled bounding box) — is rasterised
by one kernel from the boxes, so a loader only has to ship 4 integers per sample.
"""
import torch

from ._lib import call, ptr
from .config import Cfg


def scaled_bbox(xmin, ymin, xmax, ymax, orig_w, orig_h, size=None):
    """Scale + clamp exactly like new_scripy.py:541-544 (Python round(), i.e. round-half-to-even)."""
    size = Cfg.IMG_SIZE if size is None else size

    def cl(v):
        return max(0, min(size - 1, v))
    return (cl(round(xmin * size / orig_w)), cl(round(ymin * size / orig_h)),
            cl(round(xmax * size / orig_w)), cl(round(ymax * size / orig_h)))


def attn_masks(boxes_scaled, size=None, device="cuda:0"):
    """boxes_scaled: sequence of (x0, y0, x1, y1) from `scaled_bbox` -> (B, S, S) fp32 mask on `device`."""
    size = Cfg.IMG_SIZE if size is None else size
    boxes = torch.tensor(list(boxes_scaled), dtype=torch.int32).reshape(-1, 4).to(device)
    out = torch.empty((boxes.shape[0], size, size), dtype=torch.float32, device=device)
    call("dm_attn_mask", ptr(boxes), ptr(out), boxes.shape[0], size, float(Cfg.LOW_WEIGHT), float(Cfg.MID_WEIGHT), float(Cfg.HIGH_WEIGHT))
    return out
