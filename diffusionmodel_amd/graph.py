"""One hipGraph for the whole train step (zero_grad, loss forward, backward, clip + AdamW, weight re-packs).

The eager step issues ~1300 kernel launches from Python; at ~12 us of host work per launch that is within
15 % of the GPU time of the step, so the host starts to pace the device.  Every launch of the step goes to
torch's current stream (the C ABI takes the stream as an argument) and nothing in it needs the host:
the timesteps / context mask come from torch's device RNG, the DDPM noise from the Philox kernel with a
device-resident stream offset, the AdamW bias corrections from a device-resident step count.  So the step
is captured once and replayed.

    step = GraphedTrainStep(ddpm, opt, x, c, attn_mask)     # warm-up + capture; model / optimiser state is left untouched
    loss = step(x, c, attn_mask)                             # copies the batch in, replays, returns the (static) loss tensor

Constraints: fixed batch shape; single process (no collective inside the capture — use the eager step
with GradReducer for data parallelism); hyper-parameters are read from the param group between replays.
"""
import torch

from . import ops
from ._lib import DmError
from .modules import _HipBlock


class GraphedTrainStep:
    def __init__(self, ddpm, opt, x, c, attn_mask, warmup=2):
        if not x.is_cuda:
            raise DmError("GraphedTrainStep needs device tensors")
        self.ddpm, self.opt = ddpm, opt
        self.x, self.c, self.am = x.clone(), c.clone(), attn_mask.clone()
        self._specs = [sp for m in ddpm.modules() if isinstance(m, _HipBlock) for sp in m._specs()]
        dev = x.device
        if getattr(ddpm, "_rng_dev", None) is None or ddpm._rng_dev.device != dev:
            ddpm._rng_dev = torch.full((1,), ddpm._rng_calls, dtype=torch.int64, device=dev)

        # ---- everything the warm-up steps change is put back afterwards
        snap = dict(p=opt.flat_p.clone(), m=opt.exp_avg.clone(), v=opt.exp_avg_sq.clone(), t=opt._step_dev.clone(),
                    rng=ddpm._rng_dev.clone(), bufs=[b.clone() for b in ddpm.buffers()], torch_rng=torch.cuda.get_rng_state(dev),
                    step=opt._step, calls=ddpm._rng_calls, nbt=[sp.nbt_pending for sp in self._specs])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # allocator and cache warm-up off the capture stream
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        before = [sp.nbt_pending for sp in self._specs]
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._eager()
        self._nbt = [sp.nbt_pending - b for sp, b in zip(self._specs, before)]     # BatchNorm batch counters of one step
        with torch.no_grad():
            opt.flat_p.copy_(snap["p"]); opt.exp_avg.copy_(snap["m"]); opt.exp_avg_sq.copy_(snap["v"])
            opt._step_dev.copy_(snap["t"]); ddpm._rng_dev.copy_(snap["rng"])
            for b, s in zip(ddpm.buffers(), snap["bufs"]):
                b.copy_(s)
        torch.cuda.set_rng_state(snap["torch_rng"], dev)
        opt._step, ddpm._rng_calls = snap["step"], snap["calls"]
        for sp, n in zip(self._specs, snap["nbt"]):
            sp.nbt_pending = n
        opt.flat_g.zero_()
        opt.refresh_shadow()
        ops.bump_weight_epoch()
        ops.refresh_packs()
        self.replays = 0

    def _eager(self):
        self.opt.zero_grad()
        loss = self.ddpm(self.x, self.c, self.am)
        loss.backward()
        self.opt.step()
        return loss

    def __call__(self, x=None, c=None, attn_mask=None):
        if x is not None:
            self.x.copy_(x, non_blocking=True)
        if c is not None:
            self.c.copy_(c, non_blocking=True)
        if attn_mask is not None:
            self.am.copy_(attn_mask, non_blocking=True)
        self.opt.sync_hyper()                              # picks up a changed lr / weight decay (outside the graph)
        self.graph.replay()
        for sp, d in zip(self._specs, self._nbt):          # host-side bookkeeping one replay stands for
            sp.nbt_pending += d
        self.opt._step += 1
        self.ddpm._rng_calls += 1
        self.replays += 1
        # (packed weights: every pack is either refreshed by the captured optimiser step (bf16 shadow, registered transposes) or
        #  was stale at capture time and is therefore rebuilt inside the graph, so eager code that follows finds fresh contents
        #  behind the stamps the capture left)
        return self.loss
