"""Autograd boundary of the MI355X path: every differentiable piece of `CLIP.forward` (reference x_clip/x_clip.py:597-875)
is one `torch.autograd.Function` whose forward AND backward are explicit sequences of C-ABI kernel launches
(x_clip_amd.ops -> libxclip_hip.so).  torch provides tensors, streams and the autograd graph between these nodes;
no arithmetic of the hot path is left to ATen and there is no CPU fallback.

Nodes
  text_encode     TextTransformer.forward   (x_clip.py:317-338)  tokens -> [b, n+1, D]
  vision_encode   VisionTransformer.forward (x_clip.py:372-390)  image  -> [b, 1+n_keep, D]
  transformer     Transformer.forward       (x_clip.py:274-291)  the block stack on its own
  linear / l2norm / select_row              the latent projections, F.normalize, enc[:, 0]  (x_clip.py:708-715)

The block stack keeps the residual stream in the model dtype and saves, per layer, exactly the tensors its backward
reads (pre-norm outputs, packed QKV, attention output + log-sum-exp, FF1 output, LayerNorm statistics); with
`checkpoint=True` (reference `checkpoint_during_training`, x_clip.py:69-79,280-286) it keeps only the two residual
inputs of a layer and re-runs that layer's forward inside the backward.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch
from torch.autograd.function import once_differentiable

from . import ops

Tensor = torch.Tensor

# Weight-gradient GEMMs are off the critical path of the backward (only the optimiser consumes them), so they are issued on a
# side HIP stream -- one per (device, issuing stream) -- where they run beside the HBM-bound LayerNorm / attention backward
# kernels of the main chain.  Their operands (a layer's activations and gradients, gigabytes each) are NOT handed to the
# caching allocator with record_stream(): its event-deferred frees come back at timing-dependent moments, the pool fragments,
# and the step keeps calling hipMalloc (measured: ~100 device mallocs in 8 steps, 44 -> 140 GB reserved, occasional 3x slower
# steps).  Instead the operands are kept referenced until the main stream has been ordered behind the side stream's work on
# them, one layer late: `layer_done()` records an event after a layer's weight gradients and makes the main stream wait for the
# PREVIOUS layer's event before dropping that layer's operands.  Same overlap, deterministic allocation pattern.
OVERLAP_WGRAD = True
_wgrad_streams = {}

# Optional destination for parameter gradients (x_clip_amd.distributed.GradSync registers itself): an object whose
# `claim(weight, shape, dtype)` returns the buffer the gradient of the parameter stored at `weight` should be written into -- a slice of
# a persistent flat all-reduce bucket -- or None.  The backward then produces that gradient in place instead of into a fresh tensor.
# Several sinks may be registered (one GradSync per model): each is asked in turn, the first that knows the weight answers.
_grad_sinks: list = []


def add_grad_sink(sink) -> None:
    if sink not in _grad_sinks:
        _grad_sinks.append(sink)


def remove_grad_sink(sink) -> None:
    if sink in _grad_sinks:
        _grad_sinks.remove(sink)


def grad_sinks():
    return tuple(_grad_sinks)


def _grad_out(weight: Optional[Tensor], shape, dtype, device) -> Tensor:
    if weight is not None:
        for sink in _grad_sinks:
            out = sink.claim(weight, shape, dtype)
            if out is not None:
                return out
    return torch.empty(*shape, dtype=dtype, device=device)


class _SideGemm:
    def __init__(self, device):
        self.main = self.side = None
        if OVERLAP_WGRAD and device.type == "cuda":
            self.main = torch.cuda.current_stream(device)
            key = (device, self.main.cuda_stream)
            self.side = _wgrad_streams.get(key)
            if self.side is None:
                self.side = _wgrad_streams[key] = torch.cuda.Stream(device=device)
        self.keep = []                                           # operands of the current layer's side-stream GEMMs
        self.pending = []                                        # [(event, operands)] of finished layers, oldest first

    def wgrad(self, dy: Tensor, x: Tensor, N1: int, N2: int, M: int, w: Optional[Tensor] = None) -> Tensor:
        """dW [N1, N2] = dy^T x (contraction over the M token rows), issued on the side stream; `w`: the weight it is the gradient of"""
        out = _grad_out(w, (N1, N2), dy.dtype, dy.device)             # main-stream memory (or a bucket slice): consumed after join()
        if self.side is None:
            return ops.gemm(dy, x, N1, N2, M, a_kmajor=True, b_kmajor=True, out=out)
        self.side.wait_stream(self.main)                     # dy was just produced on the main stream
        with torch.cuda.stream(self.side):
            ops.gemm(dy, x, N1, N2, M, a_kmajor=True, b_kmajor=True, out=out)
        self.keep.append((dy, x))
        return out

    def wgrad_into(self, dy: Tensor, x: Tensor, N1: int, N2: int, M: int, out: Tensor) -> Tensor:
        """the same product written into a given [N1, N2] view (a row block of a weight gradient assembled from two products)"""
        if self.side is None:
            return ops.gemm(dy, x, N1, N2, M, a_kmajor=True, b_kmajor=True, out=out)
        self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            ops.gemm(dy, x, N1, N2, M, a_kmajor=True, b_kmajor=True, out=out)
        self.keep.append((dy, x))
        return out

    def layer_done(self):
        if self.side is None:
            return
        ev = torch.cuda.Event()
        ev.record(self.side)
        self.pending.append((ev, self.keep))
        self.keep = []
        while len(self.pending) > 1:                             # the side stream may lag one layer behind
            ev0, _ = self.pending.pop(0)
            self.main.wait_event(ev0)                            # everything the main stream does from here on is behind those GEMMs

    def join(self):
        if self.side is not None:
            self.main.wait_stream(self.side)
        self.keep, self.pending = [], []


def _draw_seed() -> int:
    """a fresh 62-bit seed from torch's default CPU generator (host side: no device sync)"""
    return int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64).item())


LAYER_PARAMS = 8          # attn_norm.g, to_qkv.w, to_out.w, to_out_norm.g, ff_norm.g, ff1.w, ff_inner_norm.g, ff2.w


@dataclass(frozen=True)
class StackSpec:
    depth: int
    heads: int
    dim_head: int = 64
    checkpoint: bool = False
    rotary: Optional[Tensor] = None         # rotary embedding on q, k, v: the inv_freq buffer (min(dim_head, 32) / 2 fp32), x_clip.py:155-176,221-223,311
    causal: bool = False                    # causal attention (the autoregressive text encoder), x_clip.py:231-234
    attn_dropout: float = 0.0               # dropout on the softmax probabilities (x_clip.py:212,241); 0 outside training
    ff_dropout: float = 0.0                 # dropout between the inner LayerNorm and the second Linear (x_clip.py:193-194)

    def __post_init__(self):
        # the attention kernels hold head slots of 64 features (the head-resident kernels) or 128 (wide heads: two 64-wide halves
        # through the tiled kernels).  A head narrower than its slot runs with its extra q / k / v columns zero
        # (clip.Transformer.stack_params pads to_qkv / to_out with zero rows / columns: q.k and the softmax are unchanged, the
        # padded output columns are zero and meet zero weights); only the scale dim_head^-0.5 is the head's own.
        if not 1 <= self.dim_head <= 128:
            raise NotImplementedError("x_clip_amd attention kernels hold heads of up to 128 dimensions (the reference default is 64)")
        if not (0.0 <= self.attn_dropout < 1.0 and 0.0 <= self.ff_dropout < 1.0):
            raise ValueError("dropout probabilities must lie in [0, 1)")

    @property
    def head_slot(self) -> int:
        """features per head slot in the packed qkv / attention output: 64, or 128 for heads wider than 64"""
        return 64 if self.dim_head <= 64 else 128


class _GainGrads:
    """All LayerNorm gain gradients of one backward share a flat fp32 accumulator (the LN backward kernels add into
    it with atomics) and are converted to the parameter dtype by one cast launch at the end."""

    def __init__(self, gains: Sequence[Tensor]):
        self.sizes = [g.numel() for g in gains]
        self.dtype = gains[0].dtype
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=gains[0].device)
        self.views = list(self.flat.split(self.sizes))

    def finish(self) -> List[Tensor]:
        if self.dtype == torch.float32:
            return self.views
        return list(ops.cast_from_f32(self.flat, self.dtype).split(self.sizes))


# ---- one pre-norm residual block pair (x_clip.py:285-289) --------------------------------------------------------------
def _layer_forward(x: Tensor, B: int, n: int, W: Sequence[Tensor], heads: int, mask: Optional[Tensor], rotary: Optional[Tensor] = None,
                   causal: bool = False, scale: float = 0.125, hs: int = 64, drop=(0.0, 0.0, 0)):
    g_attn, w_qkv, w_out, g_out, g_ff, w_ff1, g_inner, w_ff2 = W
    M, D = x.shape
    inner = heads * hs                                                                 # hs: features per head slot (64 | 128)
    p_attn, p_ff, seed = drop                                                          # dropout probabilities, this layer's seed
    h, m1, r1 = ops.layernorm_fwd(x, g_attn)                                           # PreNorm           :126
    qkv = ops.gemm(h, w_qkv, M, 3 * inner, D)                                          # to_qkv            :216
    if rotary is not None:
        ops.rotary_(qkv, n, rotary, head_dim=hs)                                               # q, k, v rotated   :221-223
    o, lse = ops.attention_fwd(qkv.view(B, n, 3 * inner), mask, heads, scale, causal, hs, p_attn, seed)   # scale/mask/softmax(/dropout) :217-244
    p = ops.gemm(o.view(M, inner), w_out, M, D, inner)                                 # to_out.0          :245
    x1, m2, r2, h2, m3, r3 = ops.layernorm_chain_fwd(p, g_out, x, g_ff)                # to_out.1 + skip :245,288 and PreNorm :126, one pass
    u = ops.gemm(h2, w_ff1, M, w_ff1.shape[0], D)                                      # net.0             :191
    a, m4, r4 = ops.layernorm_fwd(u, g_inner, geglu=True)                              # GEGLU + net.2     :192-193
    if p_ff > 0.0:
        ops.dropout(a, p_ff, seed + 1, out=a)                                          # net.3 Dropout     :194 (in place: `a` is only read by net.4)
    x2 = ops.gemm(a, w_ff2, M, D, a.shape[1], residual=x1)                             # net.4 + skip      :195,289
    return x2, (x, h, m1, r1, qkv, o, lse, p, m2, r2, x1, h2, m3, r3, u, a, m4, r4)


def _layer_backward(dx2: Tensor, saved, B: int, n: int, W: Sequence[Tensor], heads: int, mask: Optional[Tensor],
                    gain_acc: Sequence[Tensor], need_w: Sequence[bool], sg: "_SideGemm", rotary: Optional[Tensor] = None,
                    causal: bool = False, scale: float = 0.125, hs: int = 64, drop=(0.0, 0.0, 0), x2: Optional[Tensor] = None,
                    rowc: Optional[Tensor] = None, below=None):
    """dx2: gradient w.r.t. the layer output [M, D] -> (gradient w.r.t. the layer input, [dWqkv, dWout, dWff1, dWff2]).
    x2 = the layer's output (the next layer's saved input): with it the feed-forward block's first two backward steps run as one kernel;
    rowc = that kernel's per-row constants when the pass that produced dx2 already wrote them (ops.layernorm_bwd(ffn_stats=...));
    below = the same request for the layer BELOW this one (ops.ffn_stats_request): this layer's last kernel writes its input gradient"""
    p_attn, p_ff, seed = drop
    g_attn, w_qkv, w_out, g_out, g_ff, w_ff1, g_inner, w_ff2 = W
    dg_attn, dg_out, dg_ff, dg_inner = gain_acc
    x, h, m1, r1, qkv, o, lse, p, m2, r2, x1, h2, m3, r3, u, a, m4, r4 = saved
    M, D = x.shape
    inner = heads * hs
    F2 = w_ff1.shape[0]
    Fh = F2 // 2
    # feed-forward block
    if (x2 is not None or rowc is not None) and p_ff == 0.0 and ops.ffn_dgrad_geglu_ok(M, Fh, D, dx2.dtype):
        # net.4's input gradient and net.2's (GEGLU-LayerNorm) backward in one kernel: d a never reaches memory (csrc/kernels/gemm9.h)
        du, _ = ops.ffn_dgrad_geglu(dx2, w_ff2, u, g_inner, m4, r4, x2, x1, dg=dg_inner, rowc=rowc)
        d_ff2 = sg.wgrad(dx2, a, D, Fh, M, w_ff2) if need_w[3] else None
    else:
        da = ops.gemm(dx2, w_ff2, M, Fh, D, b_kmajor=True)
        d_ff2 = sg.wgrad(dx2, a, D, Fh, M, w_ff2) if need_w[3] else None      # (`a` is the dropped activation: what net.4 saw)
        if p_ff > 0.0:
            ops.dropout(da, p_ff, seed + 1, out=da)                                         # the same mask on the gradient
        du, _ = ops.layernorm_bwd(da, u, g_inner, m4, r4, geglu=True, dg=dg_inner)
        del da
    dh2 = ops.gemm(du, w_ff1, M, D, F2, b_kmajor=True)
    d_ff1 = sg.wgrad(du, h2, F2, D, M, w_ff1) if need_w[2] else None
    del du
    # PreNorm backward + skip, then to_out's LayerNorm backward (the attention block), one pass over the rows
    dx1, dp = ops.layernorm_chain_bwd(dh2, x1, g_ff, m3, r3, dx2, p, g_out, m2, r2, dg_ff, dg_out)
    del dh2
    do = ops.gemm(dp, w_out, M, inner, D, b_kmajor=True)
    d_out = sg.wgrad(dp, o.view(M, inner), D, inner, M, w_out) if need_w[1] else None
    del dp
    dqkv = ops.attention_bwd(qkv.view(B, n, 3 * inner), mask, o, do.view(B, n, inner), lse, heads, scale, causal, hs, p_attn, seed)
    del do
    if rotary is not None:
        ops.rotary_(dqkv.view(M, 3 * inner), n, rotary, inverse=True, head_dim=hs)      # the saved qkv is the rotated one; R^T maps its gradient back
    dh = ops.gemm(dqkv.view(M, 3 * inner), w_qkv, M, D, 3 * inner, b_kmajor=True)
    d_qkv = sg.wgrad(dqkv.view(M, 3 * inner), h, 3 * inner, D, M, w_qkv) if need_w[0] else None
    del dqkv
    dx, _ = ops.layernorm_bwd(dh, x, g_attn, m1, r1, dres=dx1, dg=dg_attn, ffn_stats=below)
    return dx, [d_qkv, d_out, d_ff1, d_ff2]


# ---- the LAST layer when only one token row per sample leaves the stack (the CLS row: x_clip.py:708 `enc_text[:, 0]`) -----------------
# Everything behind the attention is row-wise (to_out + its LayerNorm + skip, the feed-forward block, norm_out), so the rows nobody reads
# are dead in the forward, and in the backward their gradient is exactly zero: the loss reaches the layer through the pooled row alone.
# The pooled variants run that part on B rows instead of B n (the last text layer of the default CLIP: 1,024 rows instead of 263,168 --
# four large GEMMs and five row-kernel passes in the forward, five GEMMs and their weight gradients in the backward); the attention itself,
# to_qkv and the first LayerNorm stay dense (every key / value row feeds the pooled query).  The same function of the same inputs: exact in
# fp32; in bf16 the pooled rows' gradients are rounded at other points than in the dense layer (the LayerNorm backward's dx and the skip
# gradient are added as two bf16 tensors where the dense kernel fuses them in fp32; dh is the sum of two rounded products where the dense layer
# takes one contraction over 3 inner), so toggling `prune_unused_rows` moves results by bf16 ulps -- the tests hold pooled against dense to the
# bars of the dense layer against the oracle, not to bit equality.
def _pool_view(t2d: Tensor, B: int, n: int, row: int) -> Tensor:
    """rows b n + row of a contiguous [B n, W] tensor as a [B, W] view (row stride n W)"""
    Wd = t2d.shape[1]
    return t2d.view(B, n * Wd)[:, row * Wd:(row + 1) * Wd]


def _layer_forward_pooled(x: Tensor, B: int, n: int, W: Sequence[Tensor], heads: int, mask: Optional[Tensor], rotary: Optional[Tensor],
                          causal: bool, scale: float, hs: int, row: int):
    g_attn, w_qkv, w_out, g_out, g_ff, w_ff1, g_inner, w_ff2 = W
    M, D = x.shape
    inner = heads * hs
    h, m1, r1 = ops.layernorm_fwd(x, g_attn)
    xc = torch.empty(B, D, dtype=x.dtype, device=x.device)
    ops.copy_rows(_pool_view(x, B, n, row), xc)                                        # the skip connection's pooled rows
    if rotary is None or row == 0:
        # one query per head: to_qkv's first third on the pooled rows, the other two on every row, attention_pool.h (causal: the pooled
        # row sees the keys up to itself).  Rotary encoders: keys and values are rotated at their positions; the query at position 0 is
        # rotated by the angle 0 -- the identity (x_clip.py:155-176,221-223)
        q = ops.gemm(_pool_view(h, B, n, row), w_qkv[:inner], B, inner, D)
        kv = ops.gemm(h, w_qkv[inner:], M, 2 * inner, D)
        if rotary is not None:
            ops.rotary_(kv, n, rotary, head_dim=hs)
        oc, lse = ops.attention_pool_fwd(q, kv.view(B, n, 2 * inner), mask, heads, scale, hs, row + 1 if causal else None)
        qkv, o = (q, kv), oc                                                           # (the tape's qkv / o slots: the pooled forms)
    else:
        # (the rotary kernel walks whole packed qkv rows: the dense attention, then its pooled rows)
        qkv = ops.gemm(h, w_qkv, M, 3 * inner, D)
        ops.rotary_(qkv, n, rotary, head_dim=hs)
        o, lse = ops.attention_fwd(qkv.view(B, n, 3 * inner), mask, heads, scale, causal, hs, 0.0, 0)
        oc = _pool_view(o.view(M, inner), B, n, row)
    p = ops.gemm(oc, w_out, B, D, inner)                                               # from here on: B rows
    x1, m2, r2, h2, m3, r3 = ops.layernorm_chain_fwd(p, g_out, xc, g_ff)
    u = ops.gemm(h2, w_ff1, B, w_ff1.shape[0], D)
    a, m4, r4 = ops.layernorm_fwd(u, g_inner, geglu=True)
    x2 = ops.gemm(a, w_ff2, B, D, a.shape[1], residual=x1)
    return x2, (x, h, m1, r1, qkv, o, lse, p, m2, r2, x1, h2, m3, r3, u, a, m4, r4)


def _layer_backward_pooled(dx2: Tensor, saved, B: int, n: int, W: Sequence[Tensor], heads: int, mask: Optional[Tensor],
                           gain_acc: Sequence[Tensor], need_w: Sequence[bool], sg: "_SideGemm", rotary: Optional[Tensor], causal: bool,
                           scale: float, hs: int, row: int):
    """dx2 [B, D]: gradient w.r.t. the pooled rows of the layer output -> (gradient w.r.t. the layer input [B n, D], weight gradients)"""
    g_attn, w_qkv, w_out, g_out, g_ff, w_ff1, g_inner, w_ff2 = W
    dg_attn, dg_out, dg_ff, dg_inner = gain_acc
    x, h, m1, r1, qkv, o, lse, p, m2, r2, x1, h2, m3, r3, u, a, m4, r4 = saved
    M, D = x.shape
    inner = heads * hs
    F2 = w_ff1.shape[0]
    Fh = F2 // 2
    da = ops.gemm(dx2, w_ff2, B, Fh, D, b_kmajor=True)
    d_ff2 = sg.wgrad(dx2, a, D, Fh, B, w_ff2) if need_w[3] else None
    du, _ = ops.layernorm_bwd(da, u, g_inner, m4, r4, geglu=True, dg=dg_inner)
    dh2 = ops.gemm(du, w_ff1, B, D, F2, b_kmajor=True)
    d_ff1 = sg.wgrad(du, h2, F2, D, B, w_ff1) if need_w[2] else None
    dx1, dp = ops.layernorm_chain_bwd(dh2, x1, g_ff, m3, r3, dx2, p, g_out, m2, r2, dg_ff, dg_out)
    if rotary is None or row == 0:
        q, kv = qkv
        d_out = sg.wgrad(dp, o, D, inner, B, w_out) if need_w[1] else None
        doc = ops.gemm(dp, w_out, B, inner, D, b_kmajor=True)
        dq, dkv = ops.attention_pool_bwd(q, kv.view(B, n, 2 * inner), mask, o, doc, lse, heads, scale, hs, row + 1 if causal else None)
        if rotary is not None:
            ops.rotary_(dkv.view(M, 2 * inner), n, rotary, inverse=True, head_dim=hs)      # the saved kv is the rotated one; R^T maps its gradient back
        # d h = dkv W_kv on every row, + dq W_q on the pooled rows (added with the skip gradient below)
        dh = ops.gemm(dkv.view(M, 2 * inner), w_qkv[inner:], M, D, 2 * inner, b_kmajor=True)
        dhq = ops.gemm(dq, w_qkv[:inner], B, D, inner, b_kmajor=True)
        hr = _pool_view(dh, B, n, row)
        t = torch.empty(B, D, dtype=dh.dtype, device=dh.device)
        ops.copy_rows(hr, t)
        ops.copy_rows(ops.add_rows(t, dhq), hr)
        d_qkv = None
        if need_w[0]:                                          # both parts of the weight gradient into ONE tensor (a bucket slice, if claimed)
            d_qkv = _grad_out(w_qkv, (3 * inner, D), dp.dtype, dp.device)
            sg.wgrad_into(dq, _pool_view(h, B, n, row), inner, D, B, d_qkv[:inner])
            sg.wgrad_into(dkv.view(M, 2 * inner), h, 2 * inner, D, M, d_qkv[inner:])
        del dkv
    else:
        oc = _pool_view(o.view(M, inner), B, n, row)
        d_out = sg.wgrad(dp, oc, D, inner, B, w_out) if need_w[1] else None
        # the attention sees a gradient on the pooled query rows only (dK / dV of every row come from those queries)
        do = torch.zeros(M, inner, dtype=dp.dtype, device=dp.device)
        ops.gemm(dp, w_out, B, inner, D, b_kmajor=True, out=_pool_view(do, B, n, row))
        dqkv = ops.attention_bwd(qkv.view(B, n, 3 * inner), mask, o, do.view(B, n, inner), lse, heads, scale, causal, hs, 0.0, 0)
        del do
        ops.rotary_(dqkv.view(M, 3 * inner), n, rotary, inverse=True, head_dim=hs)
        dh = ops.gemm(dqkv.view(M, 3 * inner), w_qkv, M, D, 3 * inner, b_kmajor=True)
        d_qkv = sg.wgrad(dqkv.view(M, 3 * inner), h, 3 * inner, D, M, w_qkv) if need_w[0] else None
        del dqkv
    dx, _ = ops.layernorm_bwd(dh, x, g_attn, m1, r1, dg=dg_attn)
    # the skip connection carries a gradient on the pooled rows only
    dxr = _pool_view(dx, B, n, row)
    t = torch.empty(B, D, dtype=dx.dtype, device=dx.device)
    ops.copy_rows(dxr, t)
    ops.copy_rows(ops.add_rows(t, dx1), dxr)
    return dx, [d_qkv, d_out, d_ff1, d_ff2]


# ---- the whole stack: norm_in -> depth x (attention, feed-forward) -> norm_out ------------------------------------------
def can_pool(spec: StackSpec) -> bool:
    """may the last layer run on one row per sample (`pool_row`)?  Dropout masks are indexed by the element's position in the dense
    activation (the oracle rebuilds them from it), so a stack with dropout keeps the dense last layer"""
    return spec.depth >= 1 and spec.attn_dropout == 0.0 and spec.ff_dropout == 0.0


def stack_forward(x0: Tensor, B: int, n: int, spec: StackSpec, params: Sequence[Tensor], mask: Optional[Tensor],
                  out: Optional[Tensor] = None, out_group: int = 0, keep_tape: bool = True, pool_row: Optional[int] = None):
    """x0 [B*n, D] -> norm_out(...) [B*n, D] (or written into `out`, see ops.layernorm_fwd) and the backward tape.
    pool_row = r: only token row r of every sample is wanted -> [B, D] (the last layer's row-wise part runs on those rows alone)"""
    assert len(params) == 2 + LAYER_PARAMS * spec.depth
    assert pool_row is None or (can_pool(spec) and out is None and 0 <= pool_row < n)
    g_in, g_out = params[0], params[-1]
    x, m_in, r_in = ops.layernorm_fwd(x0, g_in)
    layers = []
    # dropout: one 64-bit seed per pass from torch's CPU generator (no device sync; torch.manual_seed makes it reproducible); layer l
    # uses seed + 2 l (attention) and seed + 2 l + 1 (feed-forward); the backward and a checkpointed re-run regenerate the same masks
    seed0 = _draw_seed() if (spec.attn_dropout > 0.0 or spec.ff_dropout > 0.0) else 0
    for l in range(spec.depth):
        W = params[1 + LAYER_PARAMS * l: 1 + LAYER_PARAMS * (l + 1)]
        if pool_row is not None and l == spec.depth - 1:
            x_next, saved = _layer_forward_pooled(x, B, n, W, spec.heads, mask, spec.rotary, spec.causal, spec.dim_head ** -0.5, spec.head_slot, pool_row)
        else:
            x_next, saved = _layer_forward(x, B, n, W, spec.heads, mask, spec.rotary, spec.causal, spec.dim_head ** -0.5, spec.head_slot,
                                           (spec.attn_dropout, spec.ff_dropout, seed0 + 2 * l))
        if keep_tape:
            layers.append((saved[0],) if spec.checkpoint else saved)
        x = x_next
    y, m_out, r_out = ops.layernorm_fwd(x, g_out, out=out, out_group=out_group)
    tape = (x0, m_in, r_in, layers, x, m_out, r_out, seed0, pool_row) if keep_tape else None
    return y, tape


def stack_backward(dy: Tensor, tape, B: int, n: int, spec: StackSpec, params: Sequence[Tensor], mask: Optional[Tensor],
                   need: Sequence[bool]):
    """dy [B*n, D] contiguous ([B, D] for a pooled forward) -> (dx0, grads aligned with `params` (None where not needed))"""
    x0, m_in, r_in, layers, x_last, m_out, r_out, seed0, pool_row = tape
    gains = [params[0]] + [params[1 + LAYER_PARAMS * l + k] for l in range(spec.depth) for k in (0, 3, 4, 6)] + [params[-1]]
    gg = _GainGrads(gains)
    grads: List[Optional[Tensor]] = [None] * len(params)
    sg = _SideGemm(dy.device)
    def stats_for(l: int):
        """the row constants of layer l's fused feed-forward backward can be written by the kernel that produces its output gradient -- the
        LayerNorm backward of the layer above it, or of norm_out -- whose input row is layer l's output (round 6: one pass over dOut, x2, x1
        per layer less).  Not with recompute (layer l's tape does not exist yet at that point), dropout, a pooled layer (l itself, or the one
        above it: its kernel's dx is completed on the pooled rows afterwards) or shapes the fused kernel does not take"""
        if l < 0 or l >= spec.depth or spec.checkpoint or spec.ff_dropout != 0.0 or layers[l] is None:
            return None
        if pool_row is not None and l >= spec.depth - 2:
            return None
        sv = layers[l]
        if len(sv) < 18:
            return None
        Wl = params[1 + LAYER_PARAMS * l: 1 + LAYER_PARAMS * (l + 1)]
        return ops.ffn_stats_request(Wl[7], Wl[6], sv[10], sv[16], sv[17])

    st = stats_for(spec.depth - 1) if pool_row is None else None
    dx, _ = ops.layernorm_bwd(dy, x_last, params[-1], m_out, r_out, dg=gg.views[-1], ffn_stats=st)
    x_out = x_last if pool_row is None else None             # the output of the layer being walked (= the saved input of the one above it)
    for l in reversed(range(spec.depth)):
        base = 1 + LAYER_PARAMS * l
        W = params[base: base + LAYER_PARAMS]
        saved = layers[l]
        pooled = pool_row is not None and l == spec.depth - 1
        if spec.checkpoint:                      # re-run the layer forward from its saved input
            if pooled:
                _, saved = _layer_forward_pooled(saved[0], B, n, W, spec.heads, mask, spec.rotary, spec.causal, spec.dim_head ** -0.5, spec.head_slot, pool_row)
            else:
                _, saved = _layer_forward(saved[0], B, n, W, spec.heads, mask, spec.rotary, spec.causal, spec.dim_head ** -0.5, spec.head_slot,
                                          (spec.attn_dropout, spec.ff_dropout, seed0 + 2 * l))
        need_w = [need[base + 1], need[base + 2], need[base + 5], need[base + 7]]
        if pooled:
            dx, dws = _layer_backward_pooled(dx, saved, B, n, W, spec.heads, mask, gg.views[1 + 4 * l: 5 + 4 * l], need_w, sg, spec.rotary, spec.causal,
                                             spec.dim_head ** -0.5, spec.head_slot, pool_row)
            st = None
        else:
            below = stats_for(l - 1)
            dx, dws = _layer_backward(dx, saved, B, n, W, spec.heads, mask, gg.views[1 + 4 * l: 5 + 4 * l], need_w, sg, spec.rotary, spec.causal, spec.dim_head ** -0.5, spec.head_slot,
                                      (spec.attn_dropout, spec.ff_dropout, seed0 + 2 * l), x2=x_out, rowc=None if st is None else st[5], below=below)
            st = below
        x_out = saved[0]
        layers[l] = None                         # release this layer's activations
        sg.layer_done()
        grads[base + 1], grads[base + 2], grads[base + 5], grads[base + 7] = dws
    dx0, _ = ops.layernorm_bwd(dx, x0, params[0], m_in, r_in, dg=gg.views[0])
    sg.join()
    gfin = gg.finish()
    grads[0], grads[-1] = gfin[0], gfin[-1]
    for l in range(spec.depth):
        base = 1 + LAYER_PARAMS * l
        for j, k in enumerate((0, 3, 4, 6)):
            grads[base + k] = gfin[1 + 4 * l + j]
    return dx0, [g if nd else None for g, nd in zip(grads, need)]


def _contig_grad(d: Optional[Tensor], like_shape, dtype, device) -> Tensor:
    if d is None:
        return torch.zeros(like_shape, dtype=dtype, device=device)
    return d if d.is_contiguous() else d.contiguous()


def _take_tape(ctx):
    """the activation tape of a node, exactly once: it is released while the backward walks it (gigabytes per layer), so a second
    backward through the same graph (retain_graph=True, two losses sharing a tower) cannot be served -- say so instead of failing
    on a None deep inside"""
    tape = ctx.tape
    if tape is None:
        raise RuntimeError("x_clip_amd: this encoder pass was already back-propagated and its activations were released during that "
                           "backward; run the forward again (or sum the losses before calling backward once) -- retain_graph is not supported")
    ctx.tape = None
    return tape


class _TransformerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec: StackSpec, mask: Optional[Tensor], grad_mode: bool, x: Tensor, *params: Tensor):
        B, n, D = x.shape
        # (ctx.needs_input_grad reflects the parameters' requires_grad even under torch.no_grad(): without the caller's grad mode
        #  an inference pass or a frozen tower would build and hold the whole training tape until forward returned)
        keep = grad_mode and any(ctx.needs_input_grad)
        x2 = ops._c(x).view(B * n, D)
        y, tape = stack_forward(x2, B, n, spec, params, mask, keep_tape=keep)
        ctx.save_for_backward(*params)                       # autograd's version counters then catch in-place edits before backward
        ctx.spec, ctx.mask, ctx.tape, ctx.shape = spec, mask, tape, (B, n, D)
        return y.view(B, n, D)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        B, n, D = ctx.shape
        params = ctx.saved_tensors
        need = ctx.needs_input_grad[4:]
        dy = _contig_grad(dy, (B, n, D), params[0].dtype, params[0].device).view(B * n, D)
        dx, grads = stack_backward(dy, _take_tape(ctx), B, n, ctx.spec, params, ctx.mask, need)
        return (None, None, None, dx.view(B, n, D) if ctx.needs_input_grad[3] else None, *grads)


def transformer(x: Tensor, params: Sequence[Tensor], spec: StackSpec, mask: Optional[Tensor] = None) -> Tensor:
    """Transformer.forward (x_clip.py:274-291) on [b, n, D]; mask: bool [b, n] key-padding mask or None."""
    return _TransformerFn.apply(spec, mask, torch.is_grad_enabled(), x, *params)


# ---- text encoder ---------------------------------------------------------------------------------------------------------
class _TextEncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec: StackSpec, tokens: Tensor, mask: Optional[Tensor], grad_mode: bool, pool_row: Optional[int], E: Tensor,
                P: Optional[Tensor], cls: Optional[Tensor], *stack: Tensor):
        B, n = tokens.shape
        npos = n + (1 if cls is not None else 0)
        D = E.shape[1]
        keep = grad_mode and any(ctx.needs_input_grad)
        ctx_pos_rows = P.shape[0] if P is not None else 0              # rows of the full table (its gradient keeps that shape)
        if P is not None and P.shape[0] != n:
            P = P[:n]                                                   # abs_pos_emb(arange(n))      x_clip.py:323
        x0 = ops.text_embed_fwd(tokens, E, P, cls)                      # token_emb + pos, cls concat  :320-331
        kmask = None
        if mask is not None:                                            # F.pad(mask, (1, 0), True)    :334
            kmask = mask if cls is None else torch.cat([mask.new_ones(B, 1), mask], dim=1)
            kmask = kmask.contiguous()
        y, tape = stack_forward(x0.view(B * npos, D), B, npos, spec, stack, kmask, keep_tape=keep, pool_row=pool_row)
        ctx.save_for_backward(*stack)
        ctx.spec, ctx.kmask, ctx.tape = spec, kmask, tape
        ctx.tokens, ctx.meta = tokens, (B, n, npos, D, E.shape[0], P is not None, cls is not None, E.dtype)
        ctx.pos_rows, ctx.pool_row = ctx_pos_rows, pool_row
        return y.view(B, npos, D) if pool_row is None else y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        B, n, npos, D, vocab, has_pos, has_cls, dtype = ctx.meta
        need = ctx.needs_input_grad[2:]                                  # (indices below: without the grad_mode and pool_row slots)
        if ctx.pool_row is None:
            dy = _contig_grad(dy, (B, npos, D), dtype, ctx.tokens.device).view(B * npos, D)
        else:
            dy = _contig_grad(dy, (B, D), dtype, ctx.tokens.device)
        dx0, sgrads = stack_backward(dy, _take_tape(ctx), B, npos, ctx.spec, ctx.saved_tensors, ctx.kmask, need[6:])
        dE = dP = dcls = None
        if need[3] or need[4] or need[5]:
            st = ops.sort_ids(ctx.tokens.reshape(-1), vocab) if need[3] else None    # ids ascending + their positions (sort.h)
            aE, aP, acls = ops.text_embed_bwd(dx0.view(B, npos, D), ctx.tokens, vocab, has_pos, has_cls, sorted_tokens=st)
            dE = ops.cast_from_f32(aE, dtype) if need[3] else None
            dP = ops.cast_from_f32(aP, dtype) if (need[4] and has_pos) else None
            if dP is not None and dP.shape[0] != ctx.pos_rows:           # text shorter than max_seq_len: the unused positions get zero
                full = torch.zeros(ctx.pos_rows, D, dtype=dP.dtype, device=dP.device)
                full[:dP.shape[0]].copy_(dP)
                dP = full
            dcls = ops.cast_from_f32(acls, dtype) if (need[5] and has_cls) else None
        return (None, None, None, None, None, dE, dP, dcls, *sgrads)


def text_encode(tokens: Tensor, mask: Optional[Tensor], E: Tensor, P: Optional[Tensor], cls: Optional[Tensor],
                stack: Sequence[Tensor], spec: StackSpec, pool_row: Optional[int] = None) -> Tensor:
    """TextTransformer.forward (x_clip.py:317-338): int64 tokens [b, n] (+ bool key mask [b, n]) -> [b, n+1, D];
    pool_row = r (the caller reads `[:, r]` and nothing else): -> [b, D], that row of every sample (stack_forward)."""
    if tokens.dtype != torch.int64:
        raise TypeError(f"text tokens must be int64, got {tokens.dtype}")
    if P is not None and tokens.shape[1] > P.shape[0]:
        raise IndexError(f"text length {tokens.shape[1]} exceeds max_seq_len {P.shape[0]}")
    if mask is not None and mask.dtype != torch.bool:
        mask = mask.bool()
    if pool_row is not None and not can_pool(spec):                   # (dropout: dense last layer, then the row)
        return select_row(_TextEncodeFn.apply(spec, tokens, mask, torch.is_grad_enabled(), None, E, P, cls, *stack), pool_row)
    return _TextEncodeFn.apply(spec, tokens, mask, torch.is_grad_enabled(), pool_row, E, P, cls, *stack)


# ---- vision encoder -----------------------------------------------------------------------------------------------------------
class _VisionEncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec: StackSpec, patch: int, image: Tensor, keep_idx: Optional[Tensor], grad_mode: bool, w_tok: Tensor,
                b_tok: Tensor, pos: Tensor, w_cls: Tensor, *stack: Tensor):
        B, C, H, Wd = image.shape
        D = w_tok.shape[0]
        npatch = (H // patch) * (Wd // patch)
        keep = grad_mode and any(ctx.needs_input_grad)
        if keep_idx is not None:                                        # PatchDropout keep-set        x_clip.py:140-151
            nk = keep_idx.shape[1]
            rowidx = keep_idx.reshape(-1)
        else:
            nk = npatch
            rowidx = torch.arange(npatch, dtype=torch.int32, device=image.device).repeat(B)
        patches = ops.patchify(image, patch, keep_idx)                  # Rearrange (p1 p2 c)          :357
        Kp = patches.shape[1]
        if w_tok.shape[1] != Kp:                                        # K padded to the 16-byte chunk (e.g. 588 -> 592)
            w_pad = w_tok.new_zeros(D, Kp)
            ops.copy_rows(w_tok, w_pad[:, : w_tok.shape[1]]) if w_tok.shape[1] % ops.vec(w_tok.dtype) == 0 else \
                w_pad[:, : w_tok.shape[1]].copy_(w_tok)
        else:
            w_pad = w_tok
        # Linear(patch_dim, dim) + bias + pos_emb[patch]                :358,382-383 (dropout after the add == gather of both)
        tok = ops.gemm(patches, w_pad, B * nk, D, Kp, bias=b_tok, addrows=pos, rowidx=rowidx)
        enc = torch.empty(B, 1 + nk, D, dtype=tok.dtype, device=tok.device)
        y, tape = stack_forward(tok, B, nk, spec, stack, None, out=enc.view(B * (1 + nk), D), out_group=nk, keep_tape=keep)
        pooled = ops.token_mean_fwd(enc[:, 1:])                         # Reduce('b n d -> b d', 'mean') :367
        ops.gemm(pooled, w_cls, B, D, D, out=enc[:, 0])                 # to_cls_tokens Linear + concat  :368,389-390
        ctx.save_for_backward(w_cls, *stack)
        ctx.spec, ctx.tape = spec, tape
        ctx.saved = (patches if keep else None, rowidx, pooled, w_pad)
        ctx.meta = (B, nk, D, npatch, Kp, w_tok.shape[1], w_tok.dtype)
        return enc

    @staticmethod
    @once_differentiable
    def backward(ctx, denc):
        B, nk, D, npatch, Kp, Kw, dtype = ctx.meta
        patches, rowidx, pooled, w_pad = ctx.saved
        w_cls, *stack = ctx.saved_tensors
        need = ctx.needs_input_grad[1:]                                  # (indices below: without the grad_mode slot)
        denc = _contig_grad(denc, (B, 1 + nk, D), dtype, pooled.device)
        dcls = denc[:, 0]                                               # [B, D], row stride (1+nk) D
        dpooled = ops.gemm(dcls, w_cls, B, D, D, b_kmajor=True)
        dw_cls = ops.gemm(dcls, pooled, D, D, B, a_kmajor=True, b_kmajor=True) if need[7] else None
        dy = ops.token_mean_bwd(dpooled, nk, add=denc[:, 1:])           # mean-pool backward + direct token gradients
        dtok, sgrads = stack_backward(dy.view(B * nk, D), _take_tape(ctx), B, nk, ctx.spec, stack, None, need[8:])
        dw_tok = db = dpos = None
        if need[4]:
            dw_tok = ops.gemm(dtok, patches, D, Kp, B * nk, a_kmajor=True, b_kmajor=True)
            if Kp != Kw:
                dw_tok = dw_tok[:, :Kw].contiguous()
        if need[5] or need[6]:
            acc_b = torch.zeros(D, dtype=torch.float32, device=dtok.device) if need[5] else None
            acc_p = torch.zeros(npatch, D, dtype=torch.float32, device=dtok.device) if need[6] else None
            if need[5]:
                ops.rows_scatter_add(dtok, None, None, acc_b)           # bias gradient = column sum
            if need[6]:
                ops.scatter_add_sorted(dtok, *ops.sort_ids(rowidx.to(torch.int64).reshape(-1), npatch), acc_p)
            db = ops.cast_from_f32(acc_b, dtype) if need[5] else None
            dpos = ops.cast_from_f32(acc_p, dtype) if need[6] else None
        return (None, None, None, None, None, dw_tok, db, dpos, dw_cls, *sgrads)


def vision_encode(image: Tensor, keep_idx: Optional[Tensor], patch: int, w_tok: Tensor, b_tok: Tensor, pos: Tensor,
                  w_cls: Tensor, stack: Sequence[Tensor], spec: StackSpec) -> Tensor:
    """VisionTransformer.forward (x_clip.py:372-390): image [b, c, H, W] -> [b, 1 + n_keep, D].  keep_idx: int32
    [b, n_keep] kept patch indices (the PatchDropout draw) or None to keep all patches.  The image itself receives no
    gradient."""
    if image.dtype != w_tok.dtype:
        raise TypeError(f"image dtype {image.dtype} must match the model dtype {w_tok.dtype}")
    if keep_idx is not None and keep_idx.dtype != torch.int32:
        keep_idx = keep_idx.to(torch.int32)
    return _VisionEncodeFn.apply(spec, patch, image, None if keep_idx is None else keep_idx.contiguous(), torch.is_grad_enabled(),
                                 w_tok, b_tok, pos, w_cls, *stack)


# ---- small differentiable pieces of the head -------------------------------------------------------------------------------
class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor):
        lead = x.shape[:-1]
        K = x.shape[-1]
        x2 = x if (x.dim() == 2 and x.stride(1) == 1) else ops._c(x).view(-1, K)
        M, N = x2.shape[0], w.shape[0]
        y = ops.gemm(x2, w, M, N, K)
        ctx.save_for_backward(x2, w)
        ctx.lead = lead
        return y.view(*lead, N)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        M, K = x2.shape
        N = w.shape[0]
        dy2 = ops._c(dy).view(M, N)
        dx = ops.gemm(dy2, w, M, K, N, b_kmajor=True).view(*ctx.lead, K) if ctx.needs_input_grad[0] else None
        dw = ops.gemm(dy2, x2, N, K, M, a_kmajor=True, b_kmajor=True, out=_grad_out(w, (N, K), dy2.dtype, dy2.device)) if ctx.needs_input_grad[1] else None
        return dx, dw


def linear(x: Tensor, w: Tensor) -> Tensor:
    """bias-free nn.Linear (to_text_latent / to_visual_latent, x_clip.py:556,570,713-714)"""
    return _LinearFn.apply(x, w)


class _DownsampleFn(torch.autograd.Function):
    """`downsample_image_embeds` projection (x_clip.py:560-568): depthwise 4 x 4 / stride 2 / pad 1 convolution over the square token
    grid, then the 1 x 1 convolution with bias = a GEMM with a bias row.  tokens [b, n, C] -> [b, n / 4, L]."""

    @staticmethod
    def forward(ctx, x: Tensor, w_dw: Tensor, w_pw: Tensor, b_pw: Tensor):
        x = ops._c(x)
        b, n, C = x.shape
        L = w_pw.shape[0]
        wd = ops._c(w_dw).view(C, 16)
        wp = ops._c(w_pw).view(L, C)
        y = ops.dwconv4s2_fwd(x, wd)                                           # [b, n/4, C]
        M = y.shape[0] * y.shape[1]
        z = ops.gemm(y.view(M, C), wp, M, L, C, bias=ops._c(b_pw))
        ctx.save_for_backward(x, wd, y, wp)
        ctx.meta = (b, n, C, L, w_dw.shape, w_pw.shape, b_pw.dtype)
        return z.view(b, y.shape[1], L)

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        x, wd, y, wp = ctx.saved_tensors
        b, n, C, L, dw_shape, pw_shape, bdt = ctx.meta
        M = y.shape[0] * y.shape[1]
        dz2 = ops._c(dz).view(M, L)
        dy = ops.gemm(dz2, wp, M, C, L, b_kmajor=True)                         # [M, C]
        d_pw = ops.gemm(dz2, y.view(M, C), L, C, M, a_kmajor=True, b_kmajor=True).view(pw_shape) if ctx.needs_input_grad[2] else None
        d_b = None
        if ctx.needs_input_grad[3]:
            acc = torch.zeros(L, dtype=torch.float32, device=dz.device)
            ops.rows_scatter_add(dz2, None, None, acc)                         # column sums
            d_b = acc.to(bdt)
        dwa = torch.zeros(C * 16, dtype=torch.float32, device=dz.device)
        dx = ops.dwconv4s2_bwd(dy.view(b, M // b, C), x, wd, dwa)
        d_dw = dwa.to(wd.dtype).view(dw_shape) if ctx.needs_input_grad[1] else None
        return (dx if ctx.needs_input_grad[0] else None), d_dw, d_pw, d_b


def downsample_latents(tokens: Tensor, w_dw: Tensor, w_pw: Tensor, b_pw: Tensor) -> Tensor:
    return _DownsampleFn.apply(tokens, w_dw, w_pw, b_pw)


class _L2NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor):
        y, rn = ops.l2norm_fwd(x)
        ctx.save_for_backward(y, rn)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        y, rn = ctx.saved_tensors
        return ops.l2norm_bwd(ops._c(dy), y, rn)


def l2norm(x: Tensor) -> Tensor:
    """F.normalize(dim=-1) (x_clip.py:54-55,715)"""
    return _L2NormFn.apply(x)


class _PairSimilarityFn(torch.autograd.Function):
    """einsum('b d, b d -> b'): the similarity of matched pairs, the reference's inference return (x_clip.py:744-746)"""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor):
        ctx.save_for_backward(a, b)
        return ops.rowdot(a, b)

    @staticmethod
    @once_differentiable
    def backward(ctx, ds):
        a, b = ctx.saved_tensors
        ds = ds.unsqueeze(-1)
        return (ds * b if ctx.needs_input_grad[0] else None), (ds * a if ctx.needs_input_grad[1] else None)


def pair_similarity(a: Tensor, b: Tensor) -> Tensor:
    return _PairSimilarityFn.apply(a, b)


def _pad_dim(x: Tensor, dim: int, mult: int) -> Tensor:
    n = x.shape[dim]
    if n % mult == 0:
        return x.contiguous()
    shape = list(x.shape)
    shape[dim] = (n + mult - 1) // mult * mult
    out = x.new_zeros(shape)
    out.narrow(dim, 0, n).copy_(x)
    return out


class _TokenSimilarityFn(torch.autograd.Function):
    """einsum('b t d, b i d -> b t i'): every text token against every image token of the MATCHED pair -- the reference's fine-grained
    inference return (x_clip.py:742-743).  One batched launch forward, one per gradient (xclip_gemm_batched)."""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor):
        B, t, d = a.shape
        i = b.shape[1]
        v = ops.vec(a.dtype)
        assert d % v == 0, "latent width must be a multiple of the 16-byte chunk"
        a, bp = a.contiguous(), _pad_dim(b, 1, v)               # (the product's N is a whole number of chunks: zero rows behind the image tokens)
        ctx.save_for_backward(a, b.contiguous())
        return ops.bmm(a, bp, t, bp.shape[1], d)[:, :, :i]

    @staticmethod
    @once_differentiable
    def backward(ctx, ds):
        a, b = ctx.saved_tensors
        B, t, d = a.shape
        i = b.shape[1]
        dsp = _pad_dim(ds, 2, ops.vec(a.dtype))                 # [B, t, ip], zero columns behind i
        da = ops.bmm(dsp, b, t, d, i, False, True) if ctx.needs_input_grad[0] else None          # dS b
        db = ops.bmm(dsp, a, i, d, t, True, True) if ctx.needs_input_grad[1] else None           # dS^T a
        return da, db


def token_similarity(a: Tensor, b: Tensor) -> Tensor:
    return _TokenSimilarityFn.apply(a, b)


class _SelectRowFn(torch.autograd.Function):
    """enc[:, index] as a contiguous [b, D] tensor (x_clip.py:708-709); the backward scatters into a zeroed buffer."""

    @staticmethod
    def forward(ctx, enc: Tensor, index: int):
        B, n, D = enc.shape
        out = torch.empty(B, D, dtype=enc.dtype, device=enc.device)
        ops.copy_rows(enc[:, index], out)
        ctx.meta = (B, n, D, index)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        B, n, D, index = ctx.meta
        denc = torch.zeros(B, n, D, dtype=dout.dtype, device=dout.device)
        ops.copy_rows(ops._c(dout), denc[:, index])
        return denc, None


def select_row(enc: Tensor, index: int = 0) -> Tensor:
    return _SelectRowFn.apply(enc, index)


class _PermuteRowsFn(torch.autograd.Function):
    """out[r] = x[idx[r]] for a PERMUTATION idx of the rows of x [rows, D]; the backward gathers with the inverse permutation"""

    @staticmethod
    def forward(ctx, x: Tensor, idx: Tensor, inv: Tensor):
        ctx.save_for_backward(inv)
        return ops.gather_rows(x, idx)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (inv,) = ctx.saved_tensors
        return ops.gather_rows(ops._c(dout), inv), None, None


def eos_to_front(enc: Tensor, tokens: Tensor, eos_id: int) -> Tensor:
    """the causal text encoder's pooling (x_clip.py:670-685): per sample, the encoding at the FIRST eos token moves to position 0
    and the other positions follow in their original order.  The index bookkeeping is integer work on the [b, n] token tensor;
    the rows move through xclip_gather_rows."""
    B, n, D = enc.shape
    is_eos = tokens == eos_id
    assert torch.all(torch.any(is_eos, dim=-1)), f'some of the text rows does not have the eos id {eos_id}'
    e = is_eos.float().argmax(dim=-1, keepdim=True)                           # first eos per row                  x_clip.py:675
    j = torch.arange(n, device=tokens.device)[None]
    src = torch.where(j == 0, e, torch.where(j <= e, j - 1, j))               # out position j <- source position
    base = torch.arange(B, device=tokens.device)[:, None] * n
    idx = (src + base).reshape(-1)
    inv = torch.empty_like(idx)
    inv[idx] = torch.arange(B * n, device=tokens.device)
    out = _PermuteRowsFn.apply(ops._c(enc).view(B * n, D), idx.to(torch.int32).contiguous(), inv.to(torch.int32).contiguous())
    return out.view(B, n, D)
