"""The contrastive head of `CLIP.forward` as one autograd node (reference x_clip/x_clip.py:736,750-769,812-868):
temperature, multiview pairing, optional cross-rank all-gather, similarity, InfoNCE / DCL, loss mix -- and its backward.

No logit matrix is materialised in the forward (fused MFMA similarity + online log-sum-exp kernels); the backward writes
the softmax-gradient factor G once (model dtype) and feeds it to MFMA GEMMs.  With torch.distributed initialised, rank r
computes only its row block (its texts vs. all images) and column block (its images vs. all texts) of every B x B
problem -- 1/W of the reference's redundant work -- and produces the same loss value on every rank and
the same gradients for its local latents as the reference's gather-everything formulation (distributed.py:41-54).

Closed form: with S = e^tau X Y^T, L = sum over view pairs w_p / (2B) [ sum_i (lse_j S_ij - S_ii)
+ sum_j (lse_i S'_ij - S'_jj) ];  G = dL/dS = cx exp(S - lse_x[:, None]) + cy exp(S - lse_y[None, :]) - (cx + cy) I
(diagonal of the exp terms removed under DCL);  dX = e^tau G Y,  dY = e^tau G^T X,  dtau = sum G o S.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
from torch.autograd.function import once_differentiable

from . import distributed as xdist
from . import ops

Tensor = torch.Tensor


@dataclass(frozen=True)
class ContrastiveSpec:
    dcl: bool = False                     # decoupled_contrastive_learning (x_clip.py:834-836)
    main_weight: float = 1.0              # cl_loss_weight (x_clip.py:855)
    multiview_weight: float = 0.0         # multiview_loss_weight when aug views are present, else 0 (x_clip.py:851-868)
    distributed: bool = False             # requires_all_gather (x_clip.py:591)
    group: object = None                  # process group (None = default)
    assume_equal_batch: bool = False      # skip the per-step size exchange (distributed.py:17-21)


def _acc_gemm(acc: Optional[Tensor], a: Tensor, b: Tensor, M: int, N: int, K: int, a_kmajor: bool) -> Tensor:
    """acc (+)= op(a) b with b k-major ([K, N]); the first product allocates acc"""
    if acc is None:
        return ops.gemm(a, b, M, N, K, a_kmajor=a_kmajor, b_kmajor=True)
    return ops.gemm(a, b, M, N, K, a_kmajor=a_kmajor, b_kmajor=True, residual=acc, out=acc)


class _ContrastiveFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec: ContrastiveSpec, tau: Tensor, T: Tensor, I: Tensor, Tx: Optional[Tensor], Ix: Optional[Tensor]):
        m, b, d = T.shape
        n = I.shape[0]
        dev = T.device
        extra = Tx is not None
        tau32 = tau.detach().reshape(1).float().contiguous()
        mats = {"T": [ops._c(T[v]) for v in range(m)], "I": [ops._c(I[v]) for v in range(n)]}
        if extra:
            mats["Tx"] = [ops._c(Tx[v]) for v in range(m)]
            mats["Ix"] = [ops._c(Ix[v]) for v in range(n)]
        # ---- cross-rank exchange (x_clip.py:759-769) ----
        gathered = {}
        if spec.distributed:
            sizes = [b] * xdist.dist.get_world_size(spec.group) if spec.assume_equal_batch else \
                xdist.exchange_sizes(b, dev, spec.group)
            rank = xdist.dist.get_rank(spec.group)
            off, B = sum(sizes[:rank]), sum(sizes)
            for name, views in mats.items():
                gathered[name] = xdist.GatheredViews(views, sizes, spec.group)
        else:
            sizes, rank, off, B = [b], 0, 0, b

        def kchunks(name, v):
            return gathered[name].chunks(v) if spec.distributed else [(mats[name][v], 0)]

        def waiter(name):
            return (lambda c: gathered[name].wait() if c == 1 else None) if spec.distributed else None

        # (X name, Y name, X view of pair (i, j), Y view of pair (i, j), cx, cy):  rows of X against all of Y
        groups = [("T", "I", lambda i, j: i, lambda i, j: j, 1.0, 0.0), ("Ix", "Tx", lambda i, j: j, lambda i, j: i, 1.0, 0.0)] \
            if extra else [("T", "I", lambda i, j: i, lambda i, j: j, 1.0, 1.0)]
        npairs = m * n
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        plan = []                                 # (group, i, j, coef_x, coef_y, index of lse_x, index of lse_y)
        lse_local: List[Tensor] = []
        for gi, (xn, yn, xv, yv, cx, cy) in enumerate(groups):
            for i in range(m):
                for j in range(n):
                    w = spec.main_weight if (i == 0 and j == 0) else spec.multiview_weight / max(npairs - 1, 1)
                    coef_x, coef_y = cx * w / (2.0 * B), cy * w / (2.0 * B)
                    ix = iy = -1
                    if coef_x != 0.0:
                        lse, _ = ops.simloss_chunked_fwd(mats[xn][xv(i, j)], kchunks(yn, yv(i, j)), 1.0, off, spec.dcl, coef_x, loss,
                                                         log_scale=tau32, before_chunk=waiter(yn))
                        ix = len(lse_local)
                        lse_local.append(lse)
                    if coef_y != 0.0:
                        lse, _ = ops.simloss_chunked_fwd(mats[yn][yv(i, j)], kchunks(xn, xv(i, j)), 1.0, off, spec.dcl, coef_y, loss,
                                                         log_scale=tau32, before_chunk=waiter(xn))
                        iy = len(lse_local)
                        lse_local.append(lse)
                    plan.append((gi, i, j, coef_x, coef_y, ix, iy))
        lse_all = None
        if spec.distributed:
            for g in gathered.values():
                g.wait()
            # every rank's log-sum-exp vectors (the only cross-rank state the backward needs) + the loss partials
            cap, world = max(sizes), len(sizes)
            send = torch.zeros(len(lse_local), cap, dtype=torch.float32, device=dev)
            for k, v in enumerate(lse_local):
                send[k, :b].copy_(v)
            lse_all = torch.empty(world, len(lse_local), cap, dtype=torch.float32, device=dev)
            w1 = xdist._gather_into(lse_all, send, spec.group, async_op=True)
            xdist.all_reduce_scalars(loss, spec.group)
            if w1 is not None:
                w1.wait()
        ctx.spec, ctx.plan, ctx.groups = spec, plan, groups
        ctx.mats, ctx.gathered, ctx.lse_local, ctx.lse_all = mats, gathered, lse_local, lse_all
        ctx.tau32, ctx.geom = tau32, (m, n, b, d, B, off, sizes, rank, extra, tau.dtype)
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        spec, plan, groups = ctx.spec, ctx.plan, ctx.groups
        mats, gathered, lse_local, lse_all, tau32 = ctx.mats, ctx.gathered, ctx.lse_local, ctx.lse_all, ctx.tau32
        m, n, b, d, B, off, sizes, rank, extra, tau_dtype = ctx.geom
        dev = tau32.device
        dt = mats["T"][0].dtype
        v = ops.vec(dt)
        gmul = dloss.detach().reshape(1).float().contiguous()
        dtau = torch.zeros(1, dtype=torch.float32, device=dev)
        grads = {name: [None] * len(views) for name, views in mats.items()}
        offsets = [sum(sizes[:r]) for r in range(len(sizes))]
        aligned = all(s % v == 0 for s in sizes)    # else: chunk columns would start off a 16-byte boundary inside G

        def lse_chunk(idx, r):
            return lse_all[r, idx, : sizes[r]] if spec.distributed else lse_local[idx]

        def block(qn, qv, kn, kv, a, c, lse_q_idx, lse_k_idx, want_dtau):
            """G for the local rows of mats[qn][qv] against every column chunk of kn/kv, then dQ += G K."""
            Q = mats[qn][qv]
            ldg = (B + v - 1) // v * v
            G = torch.empty(b, ldg, dtype=dt, device=dev)
            chunks = gathered[kn].chunks(kv) if spec.distributed else [(mats[kn][kv], 0)]
            order = [rank] + [r for r in range(len(sizes)) if r != rank]
            if spec.distributed and not aligned:
                # ragged per-rank batches (distributed.py:23-37): compact the peers' blocks once and treat them as one chunk
                by_rank = {r: ch for ch, r in zip(chunks, order)}
                chunks = [(torch.cat([by_rank[r][0] for r in range(len(sizes))], dim=0), 0)]
                order = [None]
            zero_q = None
            for (K, col0), r in zip(chunks, order):
                if lse_q_idx >= 0:
                    lq = lse_local[lse_q_idx]
                else:
                    zero_q = zero_q if zero_q is not None else torch.zeros(b, dtype=torch.float32, device=dev)
                    lq = zero_q
                if lse_k_idx < 0:
                    lk = torch.zeros(K.shape[0], dtype=torch.float32, device=dev)
                elif r is None:
                    lk = torch.cat([lse_chunk(lse_k_idx, q) for q in range(len(sizes))])
                else:
                    lk = lse_chunk(lse_k_idx, r)
                ops.simloss_grad(Q, K, 1.0, off - col0, spec.dcl, a, c, a + c, lq, ops._c(lk), dtau if want_dtau else None,
                                 log_scale=tau32, gmul=gmul, times_scale=True, out=G[:, col0: col0 + (K.shape[0] + v - 1) // v * v])
                grads[qn][qv] = _acc_gemm(grads[qn][qv], G[:, col0: col0 + K.shape[0]], K, b, d, K.shape[0], a_kmajor=False)
            return G

        for (gi, i, j, coef_x, coef_y, ix, iy) in plan:
            xn, yn, xv, yv, _, _ = groups[gi]
            xi, yi = xv(i, j), yv(i, j)
            # block A: local X rows vs all Y  ->  dX (and, single process, dY through G^T)
            G = block(xn, xi, yn, yi, coef_x, coef_y, ix, iy, True)
            if spec.distributed:
                # block B: local Y rows vs all X -> dY; dtau already counted by the row blocks
                block(yn, yi, xn, xi, coef_y, coef_x, iy, ix, False)
            else:
                grads[yn][yi] = _acc_gemm(grads[yn][yi], G[:, :b], mats[xn][xi], b, d, b, a_kmajor=True)
            del G
        if spec.distributed:
            xdist.all_reduce_scalars(dtau, spec.group)

        def stack(name, count):
            if name not in grads:
                return None
            out = torch.empty(count, b, d, dtype=dt, device=dev)
            for k, g in enumerate(grads[name]):
                if g is None:
                    out[k].zero_()
                else:
                    ops.copy_rows(g, out[k])
            return out

        ctx.mats = ctx.gathered = ctx.lse_local = ctx.lse_all = None
        need = ctx.needs_input_grad
        return (None, dtau.reshape(()).to(tau_dtype) if need[1] else None, stack("T", m) if need[2] else None,
                stack("I", n) if need[3] else None, stack("Tx", m) if (extra and need[4]) else None,
                stack("Ix", n) if (extra and need[5]) else None)


def contrastive_loss(tau: Tensor, text_latents: Tensor, image_latents: Tensor, text_latents_extra: Optional[Tensor],
                     image_latents_extra: Optional[Tensor], spec: ContrastiveSpec) -> Tensor:
    """text_latents [m, b, d], image_latents [n, b, d] (l2-normalised, m / n = number of text / image views) -> fp32 scalar
    loss = main_weight * L[0, 0] + multiview_weight * mean(other view pairs)   (x_clip.py:812-868, CLS mode)."""
    assert text_latents.dim() == 3 and image_latents.dim() == 3 and text_latents.shape[1:] == image_latents.shape[1:]
    assert (text_latents_extra is None) == (image_latents_extra is None)
    return _ContrastiveFn.apply(spec, tau, text_latents, image_latents, text_latents_extra, image_latents_extra)
