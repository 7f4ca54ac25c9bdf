"""Closed-form synthetic parameters and inputs.  TEST INFRASTRUCTURE.

The golden fixtures (tests/golden/) hold reference *outputs* only; weights and inputs are regenerated
on both sides from these formulas, so the fixtures stay small and nothing depends on a torch RNG
stream.  value(name, idx) = scale(name) * U(-1,1) drawn from numpy's frozen legacy MT19937 stream seeded with
crc32(name) — i.e. a fixed table per tensor name, identical wherever numpy runs.  (A first version
used sine waves; their strong correlations made train-mode BatchNorm over tiny batches
ill-conditioned — fp32-vs-fp64 noise of the reference itself reached 2e-4 — so it was replaced.)
"""
import math
import zlib

import numpy as np
import torch


def _wave(name, n, freq=0.0):
    """n values in [-1, 1], a pure function of (name, n): the frozen legacy MT19937 stream
    (numpy RandomState is stream-stable across numpy versions) seeded by crc32(name, freq)."""
    seed = zlib.crc32(f"{name}|{freq}".encode()) & 0x7FFFFFFF
    return np.random.RandomState(seed).uniform(-1.0, 1.0, n)


def synth_tensor(name, shape, dtype=torch.float32):
    n = int(np.prod(shape)) if len(shape) else 1
    w = _wave(name, n)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros((), dtype=torch.long)
    if leaf == "running_var":
        v = 0.6 + 0.4 * w * w
    elif leaf == "running_mean":
        v = 0.1 * w
    elif leaf in ("gamma_h", "gamma_w", "alpha", "beta"):
        v = 0.7 * w
    elif leaf == "bias":
        v = 0.05 * w
    elif leaf == "weight" and len(shape) == 1:          # BN / GN scale
        v = 1.0 + 0.2 * w
    elif leaf == "weight":
        fan_in = int(np.prod(shape[1:]))
        v = w * math.sqrt(4.5 / max(fan_in, 1))        # var = 1.5/fan_in: keeps activations O(1)
    else:
        v = w
    return torch.tensor(v.reshape(shape), dtype=dtype)


def synth_state(spec, prefix="", dtype=torch.float32):
    """spec: ordered name -> shape (oracle.unet_ref.*_spec).  Returns a fresh parameter dict."""
    return {prefix + k: synth_tensor(k, s, dtype) for k, s in spec.items()}


def synth_input(name, shape, scale=1.0, dtype=torch.float32):
    n = int(np.prod(shape))
    a = _wave(name, n, 0.7548776662) + 0.5 * _wave(name + "#2", n, 2.2360679775)
    return torch.tensor((scale * a / 1.1).reshape(shape), dtype=dtype)


def synth_noise(name, shape, dtype=torch.float32):
    """Deterministic N(0,1) field keyed by name."""
    n = int(np.prod(shape))
    seed = zlib.crc32(f"{name}|gauss".encode()) & 0x7FFFFFFF
    return torch.tensor(np.random.RandomState(seed).standard_normal(n).reshape(shape), dtype=dtype)


def synth_attn_mask(b, s):
    """Mask value convention of new_scripy.py:535-546: 0.5 background, 1.0 lower half, 3.0 in a box."""
    m = torch.full((b, s, s), 0.5)
    m[:, s // 2:, :] = 1.0
    for i in range(b):
        y0 = (3 + 5 * i) % (s // 2)
        x0 = (7 + 11 * i) % (s // 2)
        m[i, y0:y0 + s // 4, x0:x0 + s // 3] = 3.0
    return m
