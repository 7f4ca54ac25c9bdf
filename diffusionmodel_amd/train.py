"""The optimiser-step machinery of the reference's training loop (new_scripy.py:777-803) for one process or one process per GPU.

The reference accumulates `ACCUM_STEPS` micro-batches of `loss / ACCUM_STEPS`, then unscales, clips the summed gradient to
norm 1.0 and steps AdamW (:786, :795-803; a short tail group at the end of an epoch is flushed with the same 1/ACCUM_STEPS
weights).  Data parallelism keeps exactly that arithmetic: micro-batch m of the epoch goes to rank m % world, a group is
G = world * ceil(ACCUM_STEPS / world) micro-batches, every rank weighs its micro-batches by 1/G (1/local_accum in the loss,
1/world in the optimiser's gradient scale after the SUM all-reduce), BatchNorm statistics stay local to the micro-batch
(parallel.py), clipping acts on the reduced gradient.  With world == ACCUM_STEPS the step is the reference's step.

    eng = TrainEngine(ddpm, opt, accum_steps=Cfg.ACCUM_STEPS)         # joins the process group parallel.init_from_env() made
    for m, (x, c, am) in eng.my_batches(enumerate(loader)): ...
    loss = eng.micro_batch(x, c, am, last_in_epoch=...)                # backward + (at the end of a group) reduce + clip + AdamW

When a group is ONE local micro-batch of a fixed shape the whole step (draws, forward, backward, reduce markers, clip + AdamW)
is captured once and replayed as a launch plan (graph.GraphedTrainStep(mode="plan"); data parallel: the RCCL all-reduces run
between its segments, OverlappedGradReducer.replay); other shapes and accumulation groups run eagerly with the same reducer.
"""
import torch
import torch.distributed as dist

from ._lib import DmError
from . import parallel


def group_size(accum_steps, world):
    """(micro-batches per optimiser step over all ranks, per rank)."""
    local = max(1, -(-int(accum_steps) // int(world)))
    return local * world, local


def rank_batches(n_batches, rank, world):
    """Indices of the epoch's micro-batches this rank computes (micro-batch m -> rank m % world) and the number of LOCAL slots
    per rank: every rank walks `slots` positions so that all of them take part in every optimiser step; a rank whose last slot
    has no micro-batch behind it (short tail) joins that step with a zero gradient."""
    slots = -(-n_batches // world)
    mine = [s * world + rank if s * world + rank < n_batches else None for s in range(slots)]
    return mine, slots


class TrainEngine:
    def __init__(self, ddpm, opt, accum_steps=1, group=None, n_buckets=6, use_plan=True):
        self.ddpm, self.opt, self.group = ddpm, opt, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.G, self.local_accum = group_size(accum_steps, self.world)
        # loss weights: the reference divides every micro-batch loss by ACCUM_STEPS (:786) — also in a short tail group.  Here the
        # 1/G of a micro-batch is 1/local_accum on the loss and 1/world in the optimiser (after the SUM all-reduce); when
        # ACCUM_STEPS is not a multiple of the world size the group is rounded UP to G micro-batches (said once on stderr)
        self.loss_div = float(self.local_accum)
        if self.world > 1:
            opt.grad_scale = 1.0 / self.world
            opt._hyper_host = None
        self.reducer = parallel.OverlappedGradReducer(opt, n_buckets=n_buckets, group=group) if self.world > 1 else None
        self.use_plan = bool(use_plan) and self.local_accum == 1 and (self.world > 1 or int(accum_steps) == 1)
        self._planned, self._plan_shape = None, None
        self._in_group = 0            # local micro-batches accumulated since the last optimiser step
        self.opt_steps = 0
        # a replayed plan zeroes the flat gradient buffer at its START only (the capture cannot end with the fill: AdamW reads the
        # buffer last), so after a replay flat_g still holds that step's reduced gradient; the next EAGER group (short tail batch,
        # injected draws, an idle slot) must not accumulate on top of it
        self._stale_grad = False
        opt.zero_grad()

    # ---- the eager step pieces ------------------------------------------------------------------
    def _backward(self, loss, closing):
        """closing: this backward pass completes the rank's share of the group -> the reducer overlaps its buckets with it."""
        sc = self.ddpm.scaler
        if self.reducer is not None and closing:
            self.reducer.begin(capture=torch.cuda.is_current_stream_capturing())
        sc.scale(loss).backward()                                                   # :792

    def _fresh_grad(self):
        if self._stale_grad:
            if self._in_group:
                raise DmError("TrainEngine: a plan replay inside an open accumulation group")
            self.opt.zero_grad()
            self._stale_grad = False

    def _close_group(self, idle=False):
        sc = self.ddpm.scaler
        if idle:
            self._fresh_grad()
        if self.reducer is not None:
            if idle:                                   # no backward pass ran under the reducer for this group: nothing was overlapped
                self.reducer.begin()
            self.reducer.finish(idle=idle)
        sc.unscale_(self.opt)                                                        # :797 (the fused kernel unscales + clips, :798)
        sc.step(self.opt)                                                            # :800
        sc.update()                                                                  # :801
        if not torch.cuda.is_current_stream_capturing():
            self.opt.zero_grad()                                                     # :803 (a captured step zeroes at its start)
        self._in_group = 0
        self.opt_steps += 1

    def _body(self, st):
        """One whole optimiser step on one local micro-batch: what the launch plan captures."""
        self.opt.zero_grad()
        loss = self.ddpm(st.x, st.c, st.am) / self.loss_div
        self._backward(loss, True)
        self._close_group()
        return loss

    # ---- public ---------------------------------------------------------------------------------
    def micro_batch(self, x, c, attn_mask, last_in_epoch=False, **draws):
        """Forward + backward of one local micro-batch; closes the group (reduce, clip, AdamW) after `local_accum` of them or at
        the end of the epoch (:795).  Returns the micro-batch loss / loss_div as a device tensor (a fresh tensor every call).
        `draws` (ts=, noise=, ctx_mask=) inject the random draws (tests); injected draws always run eagerly."""
        closing = self._in_group + 1 == self.local_accum or last_in_epoch
        shape = (tuple(x.shape), tuple(attn_mask.shape))
        if self.use_plan and not draws and closing and self._in_group == 0 and (self._planned is None or shape == self._plan_shape) \
                and self.ddpm.training:
            if self._planned is None:
                self._build_plan(x, c, attn_mask)
            steps0 = self.opt_steps
            loss = self._planned(x, c, attn_mask)
            # (a plan that fell back to eager issue — shared GPU — ran _body -> _close_group, which counted and zeroed already)
            self._stale_grad = not self._planned._fallback_eager
            self.opt_steps = steps0 + 1
            return loss.detach().clone()
        self._fresh_grad()
        loss = self.ddpm(x, c, attn_mask, **draws) / self.loss_div                  # :784-786
        self._backward(loss, closing)
        self._in_group += 1
        if closing:
            self._close_group()
        return loss.detach()

    def idle_slot(self):
        """This rank has no micro-batch behind its last slot of the epoch (short tail) but its peers do: close the group with
        what it has accumulated so far (possibly nothing), so that every rank takes part in the step's collectives."""
        if self.reducer is None:
            raise DmError("idle_slot is a data-parallel notion")
        self._close_group(idle=True)

    def _build_plan(self, x, c, attn_mask):
        from .graph import GraphedTrainStep
        self._plan_shape = (tuple(x.shape), tuple(attn_mask.shape))
        # the constructor runs eager warm-up steps through _body (the reducer's observation step among them) and restores the
        # model / optimiser / RNG state afterwards; counters this object keeps are put back here
        steps0 = self.opt_steps
        self._planned = GraphedTrainStep(self.ddpm, self.opt, x, c, attn_mask.float(), mode="plan", body=self._body,
                                         runner=self.reducer.replay if self.reducer is not None else None)
        self.opt_steps = steps0
        self.opt.zero_grad()

    @property
    def plan(self):
        return None if self._planned is None else self._planned.plan


def reduce_mean_scalar(total, count, device, group=None):
    """(sum over ranks of total) / (sum over ranks of count): the validation loss of a sharded validation pass."""
    t = torch.tensor([float(total), float(count)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) != "nccl":
            t = t.cpu()
        dist.all_reduce(t, group=group)
    return float(t[0]) / max(float(t[1]), 1.0)


class StridedBatchSampler(torch.utils.data.Sampler):
    """Batch sampler of the data-parallel epoch: ONE seeded permutation of the dataset per epoch (identical on every rank), cut
    into micro-batches of `batch_size` (the last one may be short, like DataLoader(drop_last=False), new_scripy.py:700-707);
    rank r yields micro-batches r, r + world, ...  `slots` = micro-batch positions per rank (see rank_batches)."""

    def __init__(self, n, batch_size, rank=0, world=1, shuffle=True, seed=0):
        self.n, self.bs, self.rank, self.world, self.shuffle, self.seed, self.epoch = int(n), int(batch_size), rank, world, shuffle, int(seed), 0
        self.n_batches = -(-self.n // self.bs)
        self.mine, self.slots = rank_batches(self.n_batches, rank, world)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return sum(1 for m in self.mine if m is not None)

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator().manual_seed((self.seed * 1000003 + self.epoch) & 0x7FFFFFFFFFFFFFFF)
            perm = torch.randperm(self.n, generator=g).tolist()
        else:
            perm = list(range(self.n))
        for m in self.mine:
            if m is not None:
                yield perm[m * self.bs:(m + 1) * self.bs]
