"""Data-parallel exchange for the contrastive head: the replacement for reference x_clip/distributed.py.

The reference gathers the stacked latents of all ranks with list-form `torch.distributed.all_gather`, pads to the largest
per-rank batch, concatenates and strips the padding with an index_select (distributed.py:14-39), and its backward keeps
the local slice of the incoming gradient (distributed.py:51-54).  Here:

  * `all_gather(t, dim, sizes)` keeps that public contract (same argument meaning, same `(gathered, sizes)` result, same
    backward) for users who call it directly, but issues ONE flat `all_gather_into_tensor` (RCCL ncclAllGather over xGMI
    on MI355X, gloo on CPU) when all ranks hold the same size, and pads only when they do not.  The reference file does not
    run as shipped (`exists` and `F` are undefined names); this module restates its intent.
  * `GatheredViews` is what `CLIP.forward` itself uses: every latent view is gathered asynchronously (the collective runs
    on the process group's own stream), the loss kernels start on the rank's local block meanwhile and wait for the peers'
    blocks only when they reach them; nothing is concatenated or unpadded -- the kernels consume per-rank chunks in place.
  * `all_reduce_scalars` sums small fp32 vectors (loss partials, d tau) across ranks.

One process per GPU; the process group is whatever the user initialised (`nccl` = RCCL on ROCm, `gloo` in the CPU tests).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

Tensor = torch.Tensor


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def exchange_sizes(n: int, device, group=None) -> List[int]:
    """every rank's extent along the gather dimension (reference distributed.py:17-21).  One tiny collective + host read."""
    world = dist.get_world_size(group)
    mine = torch.tensor([n], dtype=torch.int64, device=device)
    out = torch.empty(world, dtype=torch.int64, device=device)
    _gather_into(out, mine, group, async_op=False)
    return [int(v) for v in out.tolist()]


def _gather_into(out: Tensor, inp: Tensor, group, async_op: bool):
    """flat all-gather of equal-size contiguous buffers: out [world * inp.numel()] <- inp"""
    try:
        return dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=group, async_op=async_op)
    except (RuntimeError, NotImplementedError):          # backend without the flat primitive: list form into views of `out`
        world = dist.get_world_size(group)
        parts = list(out.view(world, -1).unbind(0))
        return dist.all_gather(parts, inp.view(-1), group=group, async_op=async_op)


class GatheredViews:
    """All-gather of a set of equally shaped [rows, d] latent matrices, consumed per rank-chunk.

    chunks(v) -> [(tensor [rows_r, d], first global row)] with the LOCAL chunk first (available immediately) and the
    peers' chunks after it; call `wait()` before touching a peer chunk."""

    def __init__(self, views: Sequence[Tensor], sizes: Sequence[int], group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.sizes = list(sizes)
        self.offsets = [sum(self.sizes[:r]) for r in range(self.world)]
        self.total = sum(self.sizes)
        self.local = [v.contiguous() for v in views]
        rows, d = self.local[0].shape
        assert rows == self.sizes[self.rank]
        cap = max(self.sizes)
        self._works = []
        self.bufs = []
        for v in self.local:
            if cap != rows:                                    # uneven batch: pad to the largest (distributed.py:23-24)
                send = v.new_zeros(cap, d)
                send[:rows].copy_(v)
            else:
                send = v
            buf = torch.empty(self.world, cap, d, dtype=v.dtype, device=v.device)
            self._works.append(_gather_into(buf, send, group, async_op=True))
            self.bufs.append(buf)
        self._waited = False

    def wait(self):
        if not self._waited:
            for w in self._works:
                if w is not None:
                    w.wait()                                   # orders the current stream after the collective
            self._waited = True

    def chunks(self, v: int) -> List[Tuple[Tensor, int]]:
        out = [(self.local[v], self.offsets[self.rank])]
        for r in range(self.world):
            if r != self.rank:
                out.append((self.bufs[v][r, : self.sizes[r]], self.offsets[r]))
        return out


def all_reduce_scalars(t: Tensor, group=None) -> Tensor:
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


# ---- reference-compatible public function --------------------------------------------------------------------------------
class AllGather(torch.autograd.Function):
    """`all_gather(t, dim, sizes) -> (gathered, sizes)` (reference distributed.py:41-56)."""

    @staticmethod
    def forward(ctx, x: Tensor, dim: int, sizes: Optional[Tensor]):
        assert is_distributed(), "torch.distributed must be initialised with world_size > 1"
        world, rank = dist.get_world_size(), dist.get_rank()
        dim = dim if dim >= 0 else x.dim() + dim
        if sizes is None:
            size_list = exchange_sizes(x.shape[dim], x.device)
            sizes = torch.tensor(size_list, dtype=torch.long, device=x.device)
        else:
            size_list = [int(s) for s in sizes.tolist()]
        cap = max(size_list)
        xm = x.movedim(dim, 0).contiguous()                      # gather along the leading dim of a contiguous buffer
        if xm.shape[0] != cap:
            pad = xm.new_zeros((cap,) + tuple(xm.shape[1:]))
            pad[: xm.shape[0]].copy_(xm)
            xm = pad
        buf = torch.empty((world,) + tuple(xm.shape), dtype=x.dtype, device=x.device)
        _gather_into(buf, xm, None, async_op=False)
        if all(s == cap for s in size_list):
            g = buf.view((world * cap,) + tuple(xm.shape[1:]))
        else:
            g = torch.cat([buf[r, : size_list[r]] for r in range(world)], dim=0)
        ctx.size_list, ctx.dim, ctx.rank = size_list, dim, rank
        return g.movedim(0, dim), sizes

    @staticmethod
    def backward(ctx, grads, _):
        return grads.split(ctx.size_list, dim=ctx.dim)[ctx.rank], None, None


all_gather = AllGather.apply


# ---- data-parallel gradient averaging (the reference leaves this to the user's DDP wrapper; bench.py needs it) -----------
_ALIGN = 128        # bytes: every parameter's slice of a flat bucket starts on a cache line (GEMM outputs need 16)


class GradSync:
    """Bucketed gradient all-reduce overlapped with the backward, over PERSISTENT flat buffers.

    Parameters are grouped into buckets (one per large top-level sub-module: vision tower, text tower; the rest together); each bucket
    owns one flat buffer allocated once, in which every parameter has a fixed, 128-byte aligned slice.  The backward's weight-gradient
    GEMMs write straight into those slices (`claim`, reached through x_clip_amd.functional's grad sink; autograd then adopts the slice as
    `.grad` without a copy), gradients produced elsewhere (gains, embeddings, biases: a few % of the bytes) are copied into their slices
    when the bucket is launched -- no per-step `torch.cat` of the whole gradient.  When the last gradient of a bucket has been
    accumulated the bucket is all-reduced asynchronously on the process group's stream (RCCL over xGMI on MI355X) while autograd keeps
    running the remaining backward; `finish()` waits and scales by 1/world.  Equivalent to DistributedDataParallel's mean reduction.

    When is a bucket complete?  A parameter's post-accumulate hook fires once per backward pass THROUGH it, and a tower can be walked more
    than once per step (the MLM side loss encodes the masked text with the same tower, SimSiam runs the vision tower four more times,
    `image_micro_batches` slices it) -- so "every parameter has fired once" is not "the bucket is final".  The first step therefore
    reduces every bucket in `finish()` and records how many firings each bucket saw; from the second step on a bucket is launched from
    the hook that brings its count to that number (`overlap=False` keeps every launch in `finish()`: for graphs that change from step to
    step).  A firing that arrives after its bucket was launched means the graph grew since the last step: that is an error, not a
    silently wrong gradient.

    Stream ordering is explicit, not inherited: the towers' backward nodes run on different HIP streams (the vision tower on its side
    stream, weight gradients on theirs), so every post-accumulate hook records an event on the stream its gradient was accumulated on and
    the launching stream waits for all events of the bucket before the collective is enqueued (gloo would hide a missing edge here --
    its GPU collectives are host-staged and synchronous -- NCCL / RCCL would not).  Buckets always have the same byte size on every rank
    (a parameter without a gradient contributes zeros), so ranks cannot disagree about a collective's size."""

    def __init__(self, module: torch.nn.Module, group=None, overlap: bool = True):
        self.group = group
        self.world = dist.get_world_size(group)
        self.overlap = overlap
        self.buckets = []
        seen = set()                                         # a tower shared with a side-loss wrapper (mlm.transformer, visual_ssl.net)
                                                             # is listed under both children: reduce every parameter once

        def fresh(ps):
            out = [p for p in ps if p.requires_grad and id(p) not in seen]
            seen.update(id(p) for p in out)
            return out

        rest = fresh(module.parameters(recurse=False))
        for _, child in module.named_children():
            ps = fresh(child.parameters())
            if sum(p.numel() for p in ps) >= (1 << 20):
                self.buckets.append(ps)                      # a tower: its own bucket, reduced while the other tower runs
            else:
                rest += ps
        if rest:
            self.buckets.append(rest)
        # one flat buffer per (bucket, dtype, device): offsets are fixed for the life of the object
        self.flats = []                                      # per bucket: list of flat tensors
        self._slot = {}                                      # id(param) -> (bucket, flat index, offset in elements)
        self._by_ptr = {}                                    # param.data_ptr() -> param (the grad sink is asked by weight storage)
        for bi, ps in enumerate(self.buckets):
            groups = {}
            for p in ps:
                groups.setdefault((p.dtype, p.device), []).append(p)
            flats = []
            for (dtype, device), gps in groups.items():
                step = max(1, _ALIGN // gps[0].element_size())
                off = 0
                for p in gps:
                    self._slot[id(p)] = (bi, len(flats), off)
                    self._by_ptr[p.data_ptr()] = p
                    off += (p.numel() + step - 1) // step * step
                flats.append(torch.zeros(off, dtype=dtype, device=device))
            self.flats.append(flats)
        self._count = [0] * len(self.buckets)
        self._seen = [0] * len(self.buckets)
        self._expected = [None] * len(self.buckets)          # hook firings per step that complete a bucket (learned in the first step)
        self._events = [[] for _ in self.buckets]
        self._claimed = set()
        self._step_open = False
        self.stats = {"in_place": 0, "copied": 0, "unused": 0}     # of the last step: gradients written straight into their slice / copied / absent
        self._works = []
        self._handles = []
        for bi, ps in enumerate(self.buckets):
            for p in ps:
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        from . import functional as XF
        XF.set_grad_sink(self)

    # ---- the slices ----
    def _view(self, p: Tensor) -> Tensor:
        """a FRESH view object of p's slice (autograd adopts a gradient without copying only if nobody else holds that tensor object)"""
        bi, fi, off = self._slot[id(p)]
        return self.flats[bi][fi][off: off + p.numel()].view(p.shape)

    def claim(self, weight: Tensor, shape, dtype) -> Optional[Tensor]:
        """grad sink protocol: the buffer a backward kernel should write the gradient of the parameter stored at `weight` into, or None
        (unknown tensor, shape / dtype mismatch, the parameter already holds a gradient -- accumulation across backward calls or a tower
        that runs twice in one step -- in which cases autograd's own accumulation into the slice applies)"""
        p = self._by_ptr.get(weight.data_ptr())
        if p is None or tuple(p.shape) != tuple(shape) or p.dtype != dtype or p.grad is not None or id(p) in self._claimed:
            return None
        self._claimed.add(id(p))
        return self._view(p)

    # ---- hooks ----
    def _make_hook(self, bi):
        def hook(param):
            if param.is_cuda:                                # the stream this gradient was accumulated on (see the class docstring)
                self._events[bi].append(torch.cuda.current_stream(param.device).record_event())
            if self._count[bi] < 0:
                raise RuntimeError("x_clip_amd GradSync: a gradient arrived for a bucket that was already all-reduced in this step -- the "
                                   "autograd graph walks this tower more often than in the previous step.  Use GradSync(model, overlap=False) "
                                   "for graphs that change between steps.")
            self._count[bi] += 1
            if self.overlap and self._expected[bi] is not None and self._count[bi] == self._expected[bi]:
                self._launch(bi)
        return hook

    def _launch(self, bi):
        ps = self.buckets[bi]
        if not self._step_open:                              # first launch of a step: fresh statistics
            self._step_open, self.stats = True, dict.fromkeys(self.stats, 0)
        if ps[0].is_cuda:
            cur = torch.cuda.current_stream(ps[0].device)
            for ev in self._events[bi]:
                cur.wait_event(ev)
        self._events[bi] = []
        with torch.no_grad():
            for p in ps:
                v = self._view(p)
                if p.grad is None:
                    v.zero_()                                # unused this step: zeros on the wire, .grad stays None
                    self.stats["unused"] += 1
                elif p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
                    p.grad = v
                    self.stats["copied"] += 1
                else:
                    self.stats["in_place"] += 1
        for flat in self.flats[bi]:
            self._works.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat))
        self._seen[bi] = self._count[bi]
        self._count[bi] = -1                                 # launched

    def finish(self):
        """call after loss.backward(): buckets whose hook count never completed (unused parameters) are flushed too"""
        for bi in range(len(self.buckets)):
            if self._count[bi] > 0:
                self._launch(bi)
            if self._count[bi] < 0:
                self._expected[bi] = self._seen[bi]          # what completed this bucket in this step completes it in the next
            self._count[bi] = 0
        for work, flat in self._works:
            if work is not None:
                work.wait()                                  # orders the current stream behind the collective
            flat.mul_(1.0 / self.world)
        self._works = []
        self._claimed.clear()
        self._step_open = False

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
        from . import functional as XF
        if XF.grad_sink() is self:
            XF.set_grad_sink(None)
