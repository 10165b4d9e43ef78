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

import math
from dataclasses import dataclass
from typing import List, Optional

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


_TWICE = ("x_clip_amd: this loss was already back-propagated and the state of its head (latent views, log-sum-exps, arg-max maps) was "
          "released during that backward; run the forward again -- retain_graph is not supported")


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
                with xdist._span("lse_gather", lse_all, lse_all.numel() * 4):
                    w1.wait()
        ctx.spec, ctx.plan, ctx.groups = spec, plan, groups
        ctx.mats, ctx.gathered, ctx.lse_local, ctx.lse_all = mats, gathered, lse_local, lse_all
        ctx.tau32, ctx.geom = tau32, (m, n, b, d, B, off, sizes, rank, extra, tau.dtype)
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        if ctx.mats is None:
            raise RuntimeError(_TWICE)
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


# =========================================================================================================================
# similarity regularisation (x_clip.py:773-784, 872-873)
# =========================================================================================================================
class _SimRegFn(torch.autograd.Function):
    """(mse(offdiag(T T^T), offdiag(I I^T)) + mse(offdiag(Tx Tx^T), offdiag(Ix Ix^T))) / 2 over the GLOBAL batch.

    Rank r evaluates its row block -- its b texts / images against all B -- as two MFMA GEMMs per pair, one pass that forms
    D = TT^T - II^T off the diagonal and accumulates sum D^2 (xclip_simreg_diff), and, because D is symmetric, obtains the
    complete gradient of its own latents from that block alone: dT_r = (2 / N) D_r T_all, dI_r = -(2 / N) D_r I_all with
    N = B (B - 1).  The gathered latents are constants of the backward (as in the reference, whose all-gather backward keeps
    the local slice, distributed.py:51-54)."""

    @staticmethod
    def forward(ctx, spec: ContrastiveSpec, T: Tensor, I: Tensor, Tx: Tensor, Ix: Tensor):
        dev, dt = T.device, T.dtype
        b, d = T.shape
        v = ops.vec(dt)
        mats = [ops._c(x.detach()) for x in (T, I, Tx, Ix)]
        if spec.distributed:
            sizes = [b] * xdist.dist.get_world_size(spec.group) if spec.assume_equal_batch else xdist.exchange_sizes(b, dev, spec.group)
            rank = xdist.dist.get_rank(spec.group)
            off, B = sum(sizes[:rank]), sum(sizes)
            gathered = xdist.GatheredViews(mats, sizes, spec.group)
            gathered.wait()
        else:
            off, B, gathered = 0, b, None
        Bp = (B + v - 1) // v * v                              # GEMM N / K must be whole 16-byte chunks: zero rows contribute nothing
        alls = []
        for k in range(4):
            if gathered is None and Bp == B:
                alls.append(mats[k])
                continue
            full = torch.zeros(Bp, d, dtype=dt, device=dev)
            for chunk, row0 in (gathered.chunks(k) if gathered is not None else [(mats[k], 0)]):
                full[row0: row0 + chunk.shape[0]].copy_(chunk)
            alls.append(full)
        sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        Ds = []
        for x, y in ((0, 1), (2, 3)):
            S1 = ops.gemm(mats[x], alls[x], b, Bp, d)          # [b, Bp] = X_r X_all^T
            S2 = ops.gemm(mats[y], alls[y], b, Bp, d)
            Ds.append(ops.simreg_diff(S1, S2, off, sumsq))
        N = max(B * (B - 1), 1)
        loss = sumsq / (2.0 * N)
        if spec.distributed:
            xdist.all_reduce_scalars(loss, spec.group)
        ctx.Ds, ctx.alls, ctx.geom = Ds, alls, (b, d, Bp, N, dt)
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        b, d, Bp, N, dt = ctx.geom
        g = dloss.detach().to(dt)
        grads = [None] * 4
        for p, (x, y) in enumerate(((0, 1), (2, 3))):
            D = ctx.Ds[p]
            if ctx.needs_input_grad[1 + x]:
                grads[x] = ops.gemm(D, ctx.alls[x], b, d, Bp, b_kmajor=True, alpha=2.0 / N) * g
            if ctx.needs_input_grad[1 + y]:
                grads[y] = ops.gemm(D, ctx.alls[y], b, d, Bp, b_kmajor=True, alpha=-2.0 / N) * g
        return (None, *grads)


def sim_reg_loss(text_latents: Tensor, image_latents: Tensor, text_latents_extra: Tensor, image_latents_extra: Tensor,
                 spec: ContrastiveSpec) -> Tensor:
    """CLS-mode latents of ONE view, each [b, d] -> fp32 scalar (the reference's boolean mask [1, B, B] restricts it to a
    single view and its `*_extra` reshapes to extra_latent_projection=True, x_clip.py:776-784)."""
    assert text_latents.dim() == 2 and text_latents.shape == image_latents.shape == text_latents_extra.shape == image_latents_extra.shape
    return _SimRegFn.apply(spec, text_latents, image_latents, text_latents_extra, image_latents_extra)


# =========================================================================================================================
# fine-grained (FILIP) head: use_all_token_embeds = True  (x_clip.py:797-811 + the shared InfoNCE / DCL tail :821-868)
# =========================================================================================================================
_FILIP_CHUNK_BYTES = 1 << 30          # workspace bound for one chunk of token similarities / routing matrix
_FILIP_BWD_FACTOR = 1                 # ... times this for the routing matrix of the backward when the forward is fused.  Measured at
                                      # configs[3], b = 512 (profiles/r03_n_filip_bwd_chunk_sizes.log, one box): 128 MB chunks 33.6 ms per step,
                                      # 256 MB 33.0, 512 MB 32.3, 1 GB 31.9; one 4 GB chunk on another box 31.6 against 30.9 -- within the pool's
                                      # box-to-box spread: larger than 1 GB buys nothing, smaller costs launches
FILIP_FUSED = True                    # forward reductions inside the token-similarity GEMM where the shape allows (ops.filip_fused_ok)


class _FilipBlock:
    """Texts X [bx, nt, d] (+ mask) against images Y [by, ni, d]: t2i / i2t [bx, by] (fp32, temperature applied) with the arg-max
    positions the backward routes through.  The [bx*nt, yc*ni] token blocks are produced chunk by chunk by the MFMA GEMM into a
    bounded workspace and reduced by filip_reduce; nothing of size bx*by*nt*ni outlives a chunk."""

    def __init__(self, X: Tensor, mask_u8: Tensor, Y: Tensor, tau32: Tensor):
        self.X, self.Y, self.mask, self.tau32 = X, Y, mask_u8, tau32
        self.bx, self.nt, self.d = X.shape
        self.by, self.ni, _ = Y.shape
        v = ops.vec(X.dtype)
        esize = X.element_size()
        per_img = self.bx * self.nt * self.ni * esize
        self.fused = FILIP_FUSED and ops.filip_fused_ok(self.nt, self.ni, self.d, X.dtype)
        if self.fused:
            # the forward never sees a chunk (filip5.h): the backward's routing matrix P is free of the reduction kernel's register cap
            yc = max(1, min(self.by, _FILIP_BWD_FACTOR * _FILIP_CHUNK_BYTES // max(per_img, 1)))
        else:
            yc = max(1, min(self.by, _FILIP_CHUNK_BYTES // max(per_img, 1)))
            yc = max(1, min(yc, (256 * v * 8) // self.ni, 2048))   # rows the row-coalesced reduction keeps in registers (filip.h: FILIP_MAXCH)
        if yc < self.by:
            # the chunk width yc * ni is the contraction length of the backward GEMM dX = P Y: a multiple of the 64-deep K step keeps
            # it on the MFMA / LDS-DMA kernel (136 images x 98 tokens fell back to the register-staged 128^2 kernel: 1.47 ms per call)
            q = 64 // math.gcd(self.ni, 64)
            yc = max(q, yc // q * q)
        self.yc = yc
        self.ld = (yc * self.ni + v - 1) // v * v
        dev = X.device
        self.t2i = torch.empty(self.bx, self.by, dtype=torch.float32, device=dev)
        self.i2t = torch.empty(self.bx, self.by, dtype=torch.float32, device=dev)
        self.kmax = torch.empty(self.bx, self.nt, self.by, dtype=torch.int16, device=dev)
        self.tmax = torch.empty(self.bx, self.by, self.ni, dtype=torch.int16, device=dev)
        self.cnt = torch.zeros(self.bx, dtype=torch.float32, device=dev)
        self._ws = None

    def _workspace(self):
        if self._ws is None:
            self._ws = torch.empty(self.bx * self.nt, self.ld, dtype=self.X.dtype, device=self.X.device)
        return self._ws

    def _image_rows(self, y0: int, yc: int):
        """token rows of images [y0, y0 + yc) as a [cols, d] matrix whose row count is a whole number of 16-byte chunks of the
        similarity rows it produces (GEMM N / K granularity): zero rows are appended when yc * ni is not (odd batch x odd token count,
        e.g. 5 images x 9 patches, or the last partial chunk); they give zero similarity columns, which filip_reduce never reads and
        filip_route leaves zero"""
        n = yc * self.ni
        v = ops.vec(self.X.dtype)
        cols = (n + v - 1) // v * v
        Yc = self.Y[y0: y0 + yc].reshape(n, self.d)
        if cols == n:
            return Yc, n
        Yp = torch.zeros(cols, self.d, dtype=self.Y.dtype, device=self.Y.device)
        ops.copy_rows(Yc, Yp[:n])
        return Yp, cols

    def forward(self):
        if self.fused:
            # the reductions run in the epilogue of the token-similarity GEMM (filip5.h): nothing of size bx * by * nt * ni exists, only
            # 4-byte partials per (token row, 64-column block) / (token column, 128-row block); images in chunks that bound them
            per_img = max(1, ops.filip_fused_workspace_bytes(self.bx, self.nt, 1, self.ni))
            yc = max(1, min(self.by, _FILIP_CHUNK_BYTES // per_img))
            ws = torch.empty(ops.filip_fused_workspace_bytes(self.bx, self.nt, yc, self.ni), dtype=torch.uint8, device=self.X.device)
            Xc, Yc = ops._c(self.X), ops._c(self.Y)
            for y0 in range(0, self.by, yc):
                ops.filip_fused_fwd(Xc, self.mask, Yc[y0: y0 + yc], self.tau32, self.t2i, self.i2t, self.kmax, self.tmax, self.cnt, ws, y0)
            return self
        X2 = self.X.reshape(self.bx * self.nt, self.d)
        for y0 in range(0, self.by, self.yc):
            yc = min(self.yc, self.by - y0)
            S = self._workspace()
            Yc, cols = self._image_rows(y0, yc)
            ops.gemm(X2, Yc, self.bx * self.nt, cols, self.d, out=S[:, :cols])
            ops.filip_reduce(S, self.mask, self.tau32, self.t2i, self.i2t, self.kmax, self.tmax, self.cnt, self.nt, self.ni, yc, y0)
        self._ws = None
        return self

    def backward(self, g1: Tensor, g2: Tensor, want_dx: bool, want_dy: bool):
        """g1 = d loss / d t2i, g2 = d loss / d i2t ([bx, by] fp32) -> dX [bx, nt, d] | None, dY [by, ni, d] | None"""
        X2 = self.X.reshape(self.bx * self.nt, self.d)
        dX = None
        dY = torch.empty(self.by * self.ni, self.d, dtype=self.Y.dtype, device=self.Y.device) if want_dy else None
        for y0 in range(0, self.by, self.yc):
            yc = min(self.yc, self.by - y0)
            P = self._workspace()
            ops.filip_route(P, self.mask, self.tau32, g1, g2, self.kmax, self.tmax, self.cnt, self.nt, self.ni, yc, y0)
            Yc, cols = self._image_rows(y0, yc)           # (padding: zero columns of P against zero rows of Yc)
            if want_dx:                                   # dX += P Yc           (contraction over the chunk's image tokens)
                dX = _acc_gemm(dX, P[:, :cols], Yc, self.bx * self.nt, self.d, cols, a_kmajor=False)
            if want_dy:                                   # dYc = P^T X          (contraction over all text tokens)
                ops.gemm(P[:, : yc * self.ni], X2, yc * self.ni, self.d, self.bx * self.nt, a_kmajor=True, b_kmajor=True,
                         out=dY[y0 * self.ni: (y0 + yc) * self.ni])
        self._ws = None
        return (None if dX is None else dX.view(self.bx, self.nt, self.d)), (None if dY is None else dY.view(self.by, self.ni, self.d))


def _gather_rows(t: Tensor, sizes, group) -> Tensor:
    """[rows_r, ...] on every rank -> [sum rows, ...] in rank order (one flat all-gather; ragged batches padded on the wire)"""
    t2 = t.reshape(t.shape[0], -1)
    gv = xdist.GatheredViews([t2], sizes, group)
    gv.wait()
    if all(sz == sizes[0] for sz in sizes):
        out = gv.bufs[0].view(sum(sizes), t2.shape[1])
    else:
        out = torch.cat([gv.bufs[0][r, : sizes[r]] for r in range(len(sizes))], dim=0)
    return out.view(sum(sizes), *t.shape[1:])


class _FilipFn(torch.autograd.Function):
    """Rank-sharded when torch.distributed is initialised: rank r evaluates the row block (its texts x all images) -- every loss
    term is indexed by a text row in this mode -- and, in the backward, the column block (all texts x its images) again to route
    the gradient of ITS image tokens; ranks exchange token latents, masks and the per-row log-sum-exp vectors only.  (The
    reference cannot run this configuration: torch.stack of text and image latents)"""

    @staticmethod
    def forward(ctx, spec: ContrastiveSpec, tau: Tensor, T: Tensor, I: Tensor, Tx: Optional[Tensor], Ix: Optional[Tensor], mask: Tensor):
        m, b, nt, d = T.shape
        n, _, ni, _ = I.shape
        dev = T.device
        extra = Tx is not None
        tau32 = tau.detach().reshape(1).float().contiguous()
        mask_u8 = mask.reshape(m, b, nt).to(torch.uint8).contiguous()
        if spec.distributed:
            sizes = [b] * xdist.dist.get_world_size(spec.group) if spec.assume_equal_batch else xdist.exchange_sizes(b, dev, spec.group)
            rank = xdist.dist.get_rank(spec.group)
            off, B = sum(sizes[:rank]), sum(sizes)
        else:
            sizes, rank, off, B = [b], 0, 0, b

        def everyone(t):                                 # [b, ...] -> [B, ...]
            return _gather_rows(ops._c(t), sizes, spec.group) if spec.distributed else ops._c(t)

        Ts = [ops._c(T[i]) for i in range(m)]
        Is_all = [everyone(I[j]) for j in range(n)]
        Txs = [ops._c(Tx[i]) for i in range(m)] if extra else None
        Ixs_all = [everyone(Ix[j]) for j in range(n)] if extra else None
        npairs = m * n
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        blocks = []
        for i in range(m):
            for j in range(n):
                w = spec.main_weight if (i == 0 and j == 0) else spec.multiview_weight / max(npairs - 1, 1)
                coef = w / (2.0 * B)
                blk1 = _FilipBlock(Ts[i], mask_u8[i], Is_all[j], tau32).forward()
                blk2 = _FilipBlock(Txs[i], mask_u8[i], Ixs_all[j], tau32).forward() if extra else blk1
                lse1 = ops.rowlse(blk1.t2i, off, spec.dcl, coef, loss)
                lse2 = ops.rowlse(blk2.i2t, off, spec.dcl, coef, loss)
                blocks.append((i, j, blk1, blk2, coef, lse1, lse2))
        if spec.distributed:
            xdist.all_reduce_scalars(loss, spec.group)
        ctx.spec, ctx.blocks = spec, blocks
        ctx.local = (Ts, Txs, [ops._c(I[j]) for j in range(n)], [ops._c(Ix[j]) for j in range(n)] if extra else None, mask_u8)
        ctx.geom = (m, n, b, nt, ni, d, extra, tau.dtype, sizes, off, B, tau32)
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        if ctx.blocks is None:
            raise RuntimeError(_TWICE)
        spec, blocks = ctx.spec, ctx.blocks
        m, n, b, nt, ni, d, extra, tau_dtype, sizes, off, B, tau32 = ctx.geom
        Ts, Txs, Is_loc, Ixs_loc, mask_u8 = ctx.local
        dev = dloss.device
        dt = Ts[0].dtype
        gmul = dloss.detach().reshape(1).float().contiguous()
        dtau = torch.zeros(1, dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad
        acc = {"T": [None] * m, "I": [None] * n, "Tx": [None] * m, "Ix": [None] * n}

        def add(name, v, g):
            if g is not None:
                acc[name][v] = g if acc[name][v] is None else _add_rows(acc[name][v], g)

        # column blocks (distributed only): all texts x this rank's images, built once per text view / image view pair
        if spec.distributed:
            T_all = [_gather_rows(t, sizes, spec.group) for t in Ts]
            Tx_all = [_gather_rows(t, sizes, spec.group) for t in Txs] if extra else None
            mask_all = [_gather_rows(mask_u8[i], sizes, spec.group).contiguous() for i in range(m)]

        for (i, j, blk1, blk2, coef, lse1, lse2) in blocks:
            g1 = ops.rowgrad(blk1.t2i, lse1, off, spec.dcl, coef, gmul, dtau)
            g2 = ops.rowgrad(blk2.i2t, lse2, off, spec.dcl, coef, gmul, dtau)
            zeros = torch.zeros_like(g1)
            local_images = not spec.distributed
            if extra:
                dX, dY = blk1.backward(g1, zeros, need[2], need[3] and local_images)
                add("T", i, dX); add("I", j, dY)
                dX, dY = blk2.backward(zeros, g2, need[4], need[5] and local_images)
                add("Tx", i, dX); add("Ix", j, dY)
            else:
                dX, dY = blk1.backward(g1, g2, need[2], need[3] and local_images)
                add("T", i, dX); add("I", j, dY)
            if spec.distributed and (need[3] or (extra and need[5])):
                # d loss / d (this rank's image tokens) needs the gradient factors of EVERY text row against the local
                # images: recompute that column block and weight it with the owners' log-sum-exp vectors
                lse1_all = _gather_rows(lse1, sizes, spec.group).contiguous()
                lse2_all = _gather_rows(lse2, sizes, spec.group).contiguous()
                cb1 = _FilipBlock(T_all[i], mask_all[i], Is_loc[j], tau32).forward()
                cb2 = _FilipBlock(Tx_all[i], mask_all[i], Ixs_loc[j], tau32).forward() if extra else cb1
                h1 = ops.rowgrad(cb1.t2i, lse1_all, -off, spec.dcl, coef, gmul, None)
                h2 = ops.rowgrad(cb2.i2t, lse2_all, -off, spec.dcl, coef, gmul, None)
                hz = torch.zeros_like(h1)
                if extra:
                    _, dY = cb1.backward(h1, hz, False, need[3]); add("I", j, dY)
                    _, dY = cb2.backward(hz, h2, False, need[5]); add("Ix", j, dY)
                else:
                    _, dY = cb1.backward(h1, h2, False, need[3]); add("I", j, dY)
        if spec.distributed:
            xdist.all_reduce_scalars(dtau, spec.group)

        def stack(name, count, shape):
            out = torch.zeros(count, *shape, dtype=dt, device=dev)
            for k, g in enumerate(acc[name]):
                if g is not None:
                    ops.copy_rows(g.reshape(-1, d), out[k].reshape(-1, d))
            return out

        ctx.blocks = ctx.local = None
        return (None, dtau.reshape(()).to(tau_dtype) if need[1] else None,
                stack("T", m, (b, nt, d)) if need[2] else None, stack("I", n, (b, ni, d)) if need[3] else None,
                stack("Tx", m, (b, nt, d)) if (extra and need[4]) else None, stack("Ix", n, (b, ni, d)) if (extra and need[5]) else None,
                None)


def _add_rows(a: Tensor, b: Tensor) -> Tensor:
    """a + b for two equally shaped gradient blocks, through the GEMM epilogue (identity product + residual) would be wasteful;
    multiview FILIP sums at most a handful of [b, n, d] blocks, done with the row-copy kernel's accumulate twin"""
    d = a.shape[-1]
    return ops.add_rows(a.reshape(-1, d), b.reshape(-1, d)).view_as(a)


def filip_loss(tau: Tensor, text_latents: Tensor, image_latents: Tensor, text_latents_extra: Optional[Tensor],
               image_latents_extra: Optional[Tensor], text_mask: Tensor, spec: ContrastiveSpec) -> Tensor:
    """text_latents [m, b, nt, d], image_latents [n, b, ni, d] (l2-normalised per token), text_mask bool [m*b, nt] -> fp32 scalar"""
    assert text_latents.dim() == 4 and image_latents.dim() == 4
    return _FilipFn.apply(spec, tau, text_latents, image_latents, text_latents_extra, image_latents_extra, text_mask)
