"""Batch data parallelism: one process per GPU, RCCL all-reduce of the flat gradient buffer over xGMI.

The reference has no distributed code; its gradient-accumulation semantics define the target
(new_scripy.py:786, 795-803): every rank is one micro-batch with LOCAL BatchNorm statistics, the
gradients are summed and divided by the number of micro-batches, clipping acts on the averaged
gradient.  So rank == micro-batch, and `ACCUM_STEPS = world_size` on one device is the oracle.

xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU), so one big ring all-reduce is bound by a
single link; the flat buffer is therefore cut into a few large buckets that RCCL can pipeline, each
issued on a side stream as soon as the backward pass has produced it (reverse registration order),
and the optimiser waits on the side stream only right before its first kernel.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def bucket_bounds(total, n_buckets, align=1024):
    """Split [0, total) into <= n_buckets contiguous ranges with `align`-element boundaries."""
    n_buckets = max(1, min(n_buckets, max(1, total // align)))
    step = -(-total // n_buckets)
    step = -(-step // align) * align
    bounds, lo = [], 0
    while lo < total:
        hi = min(total, lo + step)
        bounds.append((lo, hi))
        lo = hi
    return bounds


class GradReducer:
    """Sum-all-reduce a flat gradient buffer in a few large buckets, last bucket first (the backward
    pass fills the buffer roughly back to front).  Works on CPU tensors with gloo (tests) and on
    device tensors with RCCL."""

    def __init__(self, flat_grad, n_buckets=4, group=None):
        self.flat = flat_grad
        self.group = group
        self.bounds = bucket_bounds(flat_grad.numel(), n_buckets)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = dist.is_initialized()
        self._works = []
        self._stream = torch.cuda.Stream() if flat_grad.is_cuda else None

    def start(self):
        """Launch the bucket all-reduces (asynchronously on a side stream for device tensors)."""
        if not self.active:
            return
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                for lo, hi in reversed(self.bounds):
                    self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            for lo, hi in reversed(self.bounds):
                self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Make the reduced gradients visible to the current stream."""
        for w in self._works:
            w.wait()
        self._works = []
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    def all_reduce(self):
        self.start()
        self.finish()


def broadcast_parameters(flat_params, src=0, group=None):
    """Identical initial weights on every rank."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
