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
class GradSync:
    """Bucketed gradient all-reduce overlapped with the backward: parameters are grouped into buckets (one per top-level
    sub-module: vision tower, text tower, head); when the last gradient of a bucket has been produced the bucket is
    flattened and all-reduced asynchronously on the process group's stream (RCCL over xGMI on MI355X) while autograd
    keeps running the remaining backward.  `finish()` waits, scales by 1/world and points each `.grad` at its slice of
    the reduced flat buffer.  Equivalent to DistributedDataParallel's mean-reduction for this model."""

    def __init__(self, module: torch.nn.Module, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
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
        self._pending = []
        self._count = [0] * len(self.buckets)
        self._works = []
        self._handles = []
        for bi, ps in enumerate(self.buckets):
            for p in ps:
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _make_hook(self, bi):
        def hook(param):
            self._count[bi] += 1
            if self._count[bi] == len(self.buckets[bi]):
                self._launch(bi)
        return hook

    def _launch(self, bi):
        ps = [p for p in self.buckets[bi] if p.grad is not None]
        if not ps:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append((work, flat, ps))

    def finish(self):
        """call after loss.backward(): buckets whose hook count never completed (unused parameters) are flushed too"""
        for bi in range(len(self.buckets)):
            if 0 < self._count[bi] < len(self.buckets[bi]):
                self._launch(bi)
            self._count[bi] = 0
        for work, flat, ps in self._works:
            work.wait()
            flat.mul_(1.0 / self.world)
            off = 0
            for p in ps:
                n = p.numel()
                p.grad = flat[off: off + n].view_as(p)
                off += n
        self._works = []

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
