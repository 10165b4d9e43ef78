"""Masked-language-model side loss of the text tower: the host mirror of reference x_clip/mlm.py (`MLM`, mlm.py:36-109).

The random token masking (mlm.py:70-94) is index bookkeeping on an int64 [b, n] tensor and stays in torch; everything that
touches activations runs in the gfx950 kernels: the shared TextTransformer encodes the masked sequence, ONLY the masked
positions are gathered and projected onto the vocabulary (xclip_gather_rows + xclip_gemm with the `to_logits` bias row --
F.cross_entropy(..., ignore_index=pad) ignores every other position, so their logits are never formed), and the softmax
cross-entropy and its gradient are xclip_cross_entropy_fwd / _bwd.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import nn
from torch.autograd.function import once_differentiable

from . import ops

Tensor = torch.Tensor


class _MlmHeadFn(torch.autograd.Function):
    """emb [b, n + 1, D] (token 0 = CLS), flat row indices of the masked positions, their labels -> mean cross-entropy (fp32)"""

    @staticmethod
    def forward(ctx, emb: Tensor, rows: Tensor, labels: Tensor, W: Tensor, bias: Tensor):
        emb = ops._c(emb)
        B, n1, D = emb.shape
        V = W.shape[0]
        dt, dev = emb.dtype, emb.device
        v = ops.vec(dt)
        Vp = (V + v - 1) // v * v                              # GEMM N: whole 16-byte chunks; the padding columns are ignored
        Wc, bc = ops._c(W), ops._c(bias)
        if Vp != V:
            Wp = torch.zeros(Vp, D, dtype=dt, device=dev)
            Wp[:V].copy_(Wc)
            bp = torch.zeros(Vp, dtype=dt, device=dev)
            bp[:V].copy_(bc)
        else:
            Wp, bp = Wc, bc
        nm = rows.numel()
        x = ops.gather_rows(emb.view(B * n1, D), rows)         # [nm, D]
        logits = ops.gemm(x, Wp, nm, Vp, D, bias=bp)           # to_logits                                   mlm.py:100
        acc = torch.zeros(1, dtype=torch.float32, device=dev)
        lse = ops.cross_entropy_fwd(logits, V, labels, acc)    # F.cross_entropy over the non-ignored rows   mlm.py:103-107
        ctx.save_for_backward(x, Wp, logits, lse, labels, rows)
        ctx.meta = (B, n1, D, V, Vp, dt, W.dtype, bias.dtype)
        return (acc / nm).reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        x, Wp, logits, lse, labels, rows = ctx.saved_tensors
        B, n1, D, V, Vp, dt, wdt, bdt = ctx.meta
        nm = rows.numel()
        dev = x.device
        g = dloss.detach().float().reshape(1).contiguous()
        dlog = ops.cross_entropy_bwd_(logits, V, labels, lse, g)            # [nm, Vp], in place
        demb = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dlog, Wp, nm, D, Vp, b_kmajor=True)
            table = torch.zeros(B * n1, D, dtype=torch.float32, device=dev)
            ops.rows_scatter_add(dx, rows, table, None)
            demb = ops.cast_from_f32(table, dt).view(B, n1, D)
        if ctx.needs_input_grad[3]:
            dW = ops.gemm(dlog, x, Vp, D, nm, a_kmajor=True, b_kmajor=True)[:V].to(wdt)
        if ctx.needs_input_grad[4]:
            acc = torch.zeros(Vp, dtype=torch.float32, device=dev)
            ops.rows_scatter_add(dlog, None, None, acc)
            db = acc[:V].to(bdt)
        return demb, None, None, dW, db


def _any_of(t: Tensor, ids) -> Tensor:
    hit = torch.zeros_like(t, dtype=torch.bool)
    for i in ids:
        hit |= t == i
    return hit


def _subset_with_prob(candidates: Tensor, prob: float) -> Tensor:
    """picks, per row, a random subset of the True positions of `candidates` [b, n]: the mlm.py:22-34 procedure (uniform scores,
    top-k over the row, slot k dropped when the running count of candidates over the first k + 1 sequence positions exceeds
    ceil(prob * #candidates)), one torch.rand((b, n)) draw"""
    b, n = candidates.shape
    dev = candidates.device
    k = math.ceil(prob * n)
    quota = (candidates.sum(dim=-1, keepdim=True) * prob).ceil()
    excess = (candidates.cumsum(dim=-1) > quota)[:, :k]
    scores = torch.rand((b, n), device=dev).masked_fill(~candidates, -1e9)
    picked = scores.topk(k, dim=-1).indices + 1                # slot 0 of the scatter target collects the dropped picks
    picked = picked.masked_fill(excess, 0)
    out = torch.zeros((b, n + 1), device=dev)
    out.scatter_(-1, picked, 1)
    return out[:, 1:].bool()


class MLM(nn.Module):
    """reference MLM (mlm.py:36-109): `transformer` is the CLIP text tower itself (shared parameters), `to_logits` the vocabulary
    head.  forward(seq int64 [b, n], mask=...) -> scalar loss."""

    def __init__(self, transformer, *, dim, num_tokens, mask_prob=0.15, replace_prob=0.9, random_token_prob=0., mask_token_id=2,
                 pad_token_id=0, mask_ignore_token_ids=[]):
        super().__init__()
        self.transformer = transformer
        self.mask_prob = mask_prob
        self.replace_prob = replace_prob
        self.num_tokens = num_tokens
        self.random_token_prob = random_token_prob
        self.pad_token_id = pad_token_id
        self.mask_token_id = mask_token_id
        self.mask_ignore_token_ids = set([*mask_ignore_token_ids, pad_token_id])
        self.to_logits = nn.Linear(dim, num_tokens)
        self.masked_override: Optional[Tuple[Tensor, Tensor]] = None   # parity harness: (masked_seq, labels) instead of the random draw

    def draw(self, seq: Tensor) -> Tuple[Tensor, Tensor]:
        """(masked_seq, labels): mlm.py:70-94 -- special / pad tokens are never chosen; labels hold the original id at the chosen
        positions and the pad id elsewhere; a chosen position shows the [mask] id with probability replace_prob (or, with
        random_token_prob > 0, a random non-special token)"""
        chosen = _subset_with_prob(~_any_of(seq, self.mask_ignore_token_ids), self.mask_prob)
        labels = seq.masked_fill(~chosen, self.pad_token_id)
        masked = seq.clone().detach()
        if self.random_token_prob > 0:
            assert self.num_tokens is not None, 'num_tokens keyword must be supplied when instantiating MLM if using random token replacement'
            swap = torch.zeros_like(seq).float().uniform_(0, 1) < self.random_token_prob
            rnd = torch.randint(0, self.num_tokens, seq.shape, device=seq.device)
            swap &= ~_any_of(rnd, self.mask_ignore_token_ids)
            masked = torch.where(swap, rnd, masked)
            chosen = chosen & ~swap
        show_mask = torch.zeros_like(seq).float().uniform_(0, 1) < self.replace_prob
        masked = masked.masked_fill(chosen & show_mask, self.mask_token_id)
        return masked, labels

    def forward(self, seq, **kwargs):
        masked, labels = self.masked_override if self.masked_override is not None else self.draw(seq)
        masked, labels = masked.to(seq.device), labels.to(seq.device)
        emb = self.transformer(masked, **kwargs)                              # [b, n + 1, dim], CLS first      mlm.py:97
        b, n = labels.shape
        assert emb.dim() == 3 and emb.shape[1] == n + 1, 'the MLM head expects the text encoder to prepend one CLS token'
        pos = (labels != self.pad_token_id).nonzero()                         # rows that F.cross_entropy does not ignore
        if pos.shape[0] == 0:
            return torch.full((), float('nan'), dtype=torch.float32, device=seq.device)   # mean over nothing, like the reference
        rows = (pos[:, 0] * (n + 1) + 1 + pos[:, 1]).to(torch.int32).contiguous()          # logits[:, 1:]: skip the CLS position
        picked = labels[pos[:, 0], pos[:, 1]].contiguous()
        return _MlmHeadFn.apply(emb, rows, picked, self.to_logits.weight, self.to_logits.bias)
