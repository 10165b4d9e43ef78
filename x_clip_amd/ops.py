"""Tensor-level wrappers over the C ABI: shape/dtype checks, output allocation (torch is the allocator and the
stream provider -- plumbing only), one library call each.  No arithmetic happens in Python."""
from __future__ import annotations

from typing import Optional, Tuple

import math

import torch

from . import _lib

Tensor = torch.Tensor

_DTYPES = {torch.float32: 0, torch.bfloat16: 1}
_U64 = (1 << 64) - 1


def dtype_code(t: Tensor) -> int:
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise TypeError(f"x_clip_amd kernels support float32 and bfloat16, got {t.dtype}") from None


def vec(dtype) -> int:
    return 8 if dtype == torch.bfloat16 else 4


def ln_eps(dtype) -> float:
    """reference LayerNorm.forward: eps = 1e-5 for fp32, 1e-3 otherwise (x_clip.py:118)"""
    return 1e-5 if dtype == torch.float32 else 1e-3


def _dev_check(*ts: Optional[Tensor]):
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda and not _lib.is_emulator():
            raise RuntimeError("x_clip_amd: tensors must live on an MI355X (cuda/hip device); there is no CPU path")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"x_clip_amd: tensors on different devices ({dev} vs {t.device})")


def _stream(t: Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


def _ptr(t: Optional[Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _c(t: Tensor) -> Tensor:
    return t if t.is_contiguous() else t.contiguous()


_workspaces = {}


def workspace(device, nbytes: int) -> Optional[Tensor]:
    """grow-only scratch per (device, stream): kernels using it are ordered on that stream, and two streams (the vision tower
    runs beside the text tower) never share a scratch buffer"""
    if nbytes <= 0:
        return None
    key = (device, torch.cuda.current_stream(device).cuda_stream) if device.type == "cuda" else (device, 0)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


# ---- LayerNorm family ---------------------------------------------------------------------------------------------
def layernorm_fwd(x: Tensor, g: Tensor, res: Optional[Tensor] = None, geglu: bool = False, out: Optional[Tensor] = None,
                  out_group: int = 0) -> Tuple[Tensor, Tensor, Tensor]:
    """x: [..., D] (or [..., 2D] with geglu) -> y [..., D], mean [rows], rstd [rows].
    `out` (+ `out_group` = n): write row r of the result to row r + r // n + 1 of the 2-D buffer `out`, i.e. behind
    the CLS slot of a [b, 1+n, D] encoder output viewed as [b * (1+n), D]."""
    _dev_check(x, g, res, out)
    x = _c(x)
    width = x.shape[-1]
    dim = width // 2 if geglu else width
    rows = x.numel() // width
    if out is None:
        assert out_group == 0
        y = torch.empty(*x.shape[:-1], dim, dtype=x.dtype, device=x.device)
        ldy = dim
    else:
        y = out
        assert y.dim() == 2 and y.stride(1) == 1 and y.shape[1] == dim and y.dtype == x.dtype
        assert y.shape[0] >= rows + (rows // out_group if out_group else 0)
        ldy = y.stride(0)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    if res is not None:
        res = _c(res)
        assert res.numel() == rows * dim and res.dtype == x.dtype
    assert g.dtype == x.dtype and g.numel() == dim
    L = _lib.lib()
    probe = _probe(x)
    ev0 = probe.begin(x) if probe is not None else None
    _lib.check(L.xclip_layernorm_fwd(x.data_ptr(), width, _c(g).data_ptr(), _ptr(res), y.data_ptr(), ldy, out_group,
                                     mean.data_ptr(), rstd.data_ptr(), rows, dim, ln_eps(x.dtype), int(geglu), dtype_code(x),
                                     _stream(x)), "xclip_layernorm_fwd")
    if probe is not None:      # reads the row (+ the residual), writes the normalised row
        probe.end(x, ev0, "layernorm", 0.0, rows * (width + dim + (dim if res is not None else 0)) * x.element_size(), "ln_geglu_fwd" if geglu else "ln_fwd")
    return y, mean, rstd


def layernorm_bwd(dy: Tensor, x: Tensor, g: Tensor, mean: Tensor, rstd: Tensor, geglu: bool = False,
                  dres: Optional[Tensor] = None, dg: Optional[Tensor] = None, ffn_stats=None) -> Tuple[Tensor, Tensor]:
    """-> dx (shape of x; + dres when given), dg (fp32 [D]; accumulated into `dg` when passed).
    ffn_stats = (x1_below, wg, mean4, rstd4, inv_f, rowc) (FfnStats below; bf16, not with geglu): the kernel also writes the four per-row
    constants of the feed-forward block BELOW this LayerNorm -- whose output is `x` and whose output gradient is the dx written here --
    into rowc [rows, 4] for ffn_dgrad_geglu(rowc=...) (xclip_layernorm_bwd_ffnstats)"""
    _dev_check(dy, x, g, dres)
    dy, x = _c(dy), _c(x)
    width = x.shape[-1]
    dim = width // 2 if geglu else width
    rows = x.numel() // width
    dx = torch.empty_like(x)
    if dg is None:
        dg = torch.zeros(dim, dtype=torch.float32, device=x.device)
    if dres is not None:
        dres = _c(dres)
        assert dres.numel() == rows * dim and dres.dtype == x.dtype
    L = _lib.lib()
    ws = workspace(x.device, L.xclip_layernorm_bwd_workspace_bytes(rows, dim))
    probe = _probe(x)
    ev0 = probe.begin(x) if probe is not None else None
    if ffn_stats is not None:
        x1b, wg, mean4, rstd4, inv_f, rowc = ffn_stats
        assert not geglu and x.dtype == torch.bfloat16 and tuple(x1b.shape) == (rows, dim) and x1b.stride(-1) == 1 and x1b.dtype == x.dtype
        assert wg.dtype == torch.float32 and wg.numel() >= dim and tuple(rowc.shape) == (rows, 4) and rowc.dtype == torch.float32 and rowc.is_contiguous()
        _dev_check(x1b, wg, mean4, rstd4, rowc)
        _lib.check(L.xclip_layernorm_bwd_ffnstats(dy.data_ptr(), x.data_ptr(), width, _c(g).data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                  _ptr(dres), dx.data_ptr(), width, dg.data_ptr(), _ptr(ws), 0 if ws is None else ws.numel(),
                                                  rows, dim, x1b.data_ptr(), x1b.stride(0), wg.data_ptr(), mean4.data_ptr(), rstd4.data_ptr(),
                                                  float(inv_f), rowc.data_ptr(), dtype_code(x), _stream(x)), "xclip_layernorm_bwd_ffnstats")
    else:
        _lib.check(L.xclip_layernorm_bwd(dy.data_ptr(), x.data_ptr(), width, _c(g).data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                         _ptr(dres), dx.data_ptr(), width, dg.data_ptr(), _ptr(ws), 0 if ws is None else ws.numel(),
                                         rows, dim, int(geglu), dtype_code(x), _stream(x)), "xclip_layernorm_bwd")
    if probe is not None:      # reads dy, x (+ dres, + the lower block's input), writes dx
        probe.end(x, ev0, "layernorm", 0.0, rows * (dim + 2 * width + (dim if dres is not None else 0) + (dim if ffn_stats is not None else 0)) * x.element_size(),
                  "ln_geglu_bwd" if geglu else "ln_bwd")
    return dx, dg


FUSE_FFN_DGRAD = True      # the FF2 input gradient with the GEGLU-LayerNorm backward in its epilogue (csrc/kernels/gemm9.h); False = the two-kernel path
# Error term of the fused path (tests/kernel_cases.py case_ffn_dgrad_geglu(resid_scale=...)): its second row statistic is
# s2 = dOut . (x2 - x1) with x2 = bf16(x1 + y) -- the block's own output y recovered from the bf16 residual stream.  With R = |x1| / |y| that
# difference knows y to R 2^-9 of its scale per element, s2 / F moves by ~ R 2^-9 sqrt(D) / F |dOut| |y|, and the largest element of d(u | t)
# by ~ 30 x that relative to its scale: default shape (D = 512, F = 2048) 0.06 % x R -- below a bf16 ulp (0.4 %) up to R ~ 6, 1 % at R = 16,
# 6 % at R = 100 (measured on the emulator at D = 128, F = 256, where the term is 4 x larger: 4.2 % at R = 16; the two-kernel path 0.43 %).
# Random-init and early training have R ~ 1 (every parity fixture, the oracle tests); a model whose residual stream has grown far past its
# blocks' outputs should run with FUSE_FFN_DGRAD = False (the two-kernel path reads the bf16-rounded d a instead and has no such term).


def ffn_dgrad_geglu_ok(M: int, F: int, D: int, dtype) -> bool:
    return FUSE_FFN_DGRAD and dtype == torch.bfloat16 and bool(_lib.lib().xclip_ffn_dgrad_geglu_ok(M, F, D, 1))


FUSE_FFN_ROWSTATS = True   # round 6: the fused feed-forward backward's row pass inside the LayerNorm backward that writes its dout (functional.stack_backward)


def ffn_stats_request(w2: Tensor, g: Tensor, x1: Tensor, mean4: Tensor, rstd4: Tensor):
    """what layernorm_bwd(ffn_stats=...) needs to produce the row constants of the feed-forward block (w2 [D, F] = net.4's weight, g = net.2's
    gain, x1 [M, D] = the block's input, mean4 / rstd4 = net.2's saved statistics) -> (x1, wg, mean4, rstd4, 1 / F, rowc [M, 4] fp32 to be
    filled), or None when the fused backward does not take the shape"""
    M, D = x1.shape
    F = w2.shape[1]
    if not (FUSE_FFN_ROWSTATS and ffn_dgrad_geglu_ok(M, F, D, x1.dtype) and x1.stride(-1) == 1 and D <= 4096):
        return None
    _dev_check(w2, g, x1, mean4, rstd4)
    wg = torch.empty((D + 3) // 4 * 4, dtype=torch.float32, device=x1.device)
    _lib.check(_lib.lib().xclip_ffn_wgamma(w2.data_ptr(), w2.stride(0), _c(g).data_ptr(), wg.data_ptr(), D, F, dtype_code(x1), _stream(x1)),
               "xclip_ffn_wgamma")
    return (x1, wg, mean4, rstd4, 1.0 / F, torch.empty(M, 4, dtype=torch.float32, device=x1.device))


def ffn_dgrad_geglu(dout: Tensor, w2: Tensor, x: Tensor, g: Tensor, mean: Tensor, rstd: Tensor, x2: Optional[Tensor], x1: Optional[Tensor],
                    dg: Optional[Tensor] = None, rowc: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """d(u | t) of the feed-forward block from the gradient `dout` [M, D] of its output: the product dout w2 ([D, F] = net.4's weight) and
    the GEGLU-LayerNorm backward in ONE kernel (gemm9.h).  x [M, 2F] = net.0's output, g / mean / rstd = net.2's gain and saved statistics,
    x1 / x2 [M, D] = the block's input / output (x2 = x1 + net.4(...): the second row statistic is dout . (x2 - x1)).
    rowc [M, 4] fp32 (optional): the rows' constants as layernorm_bwd(ffn_stats=...) wrote them while producing `dout` -- x2 / x1 are then not read.
    -> (dx [M, 2F], dg fp32 [F], accumulated into `dg` when passed)"""
    M, D = dout.shape
    F = w2.shape[1]
    if rowc is not None:
        _dev_check(dout, w2, x, g, rowc)
        assert tuple(w2.shape) == (D, F) and tuple(x.shape) == (M, 2 * F) and tuple(rowc.shape) == (M, 4) and rowc.dtype == torch.float32 and rowc.is_contiguous()
        assert all(t.stride(-1) == 1 for t in (dout, w2, x))
        dx = torch.empty_like(x)
        if dg is None:
            dg = torch.zeros(F, dtype=torch.float32, device=x.device)
        L = _lib.lib()
        ws = workspace(x.device, L.xclip_ffn_dgrad_geglu_workspace_bytes(M, F, D))
        probe = _probe(x)
        ev0 = probe.begin(x, "fused_ffn_bwd") if probe is not None else None
        _lib.check(L.xclip_ffn_dgrad_geglu_rowc(dout.data_ptr(), dout.stride(0), w2.data_ptr(), w2.stride(0), x.data_ptr(), x.stride(0), _c(g).data_ptr(),
                                                rowc.data_ptr(), dx.data_ptr(), dx.stride(0), dg.data_ptr(), ws.data_ptr(), ws.numel(), M, F, D,
                                                dtype_code(x), _stream(x)), "xclip_ffn_dgrad_geglu_rowc")
        if probe is not None:      # the product's flops; reads dout, w2, x, writes dx
            probe.end(x, ev0, "fused_ffn_bwd", 2.0 * M * F * D, (M * D + D * F + 4 * M * F) * 2, (M, F, D, "NN+geglu_ln_bwd", False))
        return dx, dg
    _dev_check(dout, w2, x, g, x2, x1)
    assert tuple(w2.shape) == (D, F) and tuple(x.shape) == (M, 2 * F) and tuple(x2.shape) == (M, D) and tuple(x1.shape) == (M, D)
    assert all(t.stride(-1) == 1 for t in (dout, w2, x, x2, x1)) and mean.dtype == torch.float32 and rstd.dtype == torch.float32
    dx = torch.empty_like(x)
    if dg is None:
        dg = torch.zeros(F, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    ws = workspace(x.device, L.xclip_ffn_dgrad_geglu_workspace_bytes(M, F, D))
    probe = _probe(x)
    ev0 = probe.begin(x, "fused_ffn_bwd") if probe is not None else None
    _lib.check(L.xclip_ffn_dgrad_geglu(dout.data_ptr(), dout.stride(0), w2.data_ptr(), w2.stride(0), x.data_ptr(), x.stride(0), _c(g).data_ptr(),
                                       mean.data_ptr(), rstd.data_ptr(), x2.data_ptr(), x2.stride(0), x1.data_ptr(), x1.stride(0), dx.data_ptr(),
                                       dx.stride(0), dg.data_ptr(), ws.data_ptr(), ws.numel(), M, F, D, dtype_code(x), _stream(x)),
               "xclip_ffn_dgrad_geglu")
    if probe is not None:      # the product's flops; reads dout, w2, x, x1, x2, writes dx
        probe.end(x, ev0, "fused_ffn_bwd", 2.0 * M * F * D, (M * D * 3 + D * F + 4 * M * F) * 2, (M, F, D, "NN+geglu_ln_bwd", False))
    return dx, dg


def l2norm_fwd(x: Tensor) -> Tuple[Tensor, Tensor]:
    _dev_check(x)
    x = _c(x)
    dim = x.shape[-1]
    rows = x.numel() // dim
    y = torch.empty_like(x)
    rn = torch.empty(rows, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().xclip_l2norm_fwd(x.data_ptr(), y.data_ptr(), rn.data_ptr(), rows, dim, dtype_code(x), _stream(x)),
               "xclip_l2norm_fwd")
    return y, rn


def l2norm_bwd(dy: Tensor, y: Tensor, rn: Tensor) -> Tensor:
    _dev_check(dy, y)
    dy = _c(dy)
    dim = y.shape[-1]
    dx = torch.empty_like(y)
    _lib.check(_lib.lib().xclip_l2norm_bwd(dy.data_ptr(), y.data_ptr(), rn.data_ptr(), dx.data_ptr(), y.numel() // dim, dim,
                                           dtype_code(y), _stream(y)), "xclip_l2norm_bwd")
    return dx


# ---- embeddings / patches ---------------------------------------------------------------------------------------------
class _TokenFlag:
    """Out-of-range token ids, reported without a host sync: the embedding kernel sets a device flag (and writes NaN rows, so the
    step's loss is already loudly wrong); after each launch the flag is copied to pinned host memory behind an event, and the next
    natural point that finds that event completed -- the backward of the same encoder pass (`text_embed_bwd`), the next forward, or
    an explicit `ops.check_tokens()` (which waits for it: call it after an inference batch, or once per step next to the optimizer)
    -- raises the IndexError the reference's nn.Embedding raises (x_clip.py:320), naming the launch that saw the bad id.  One flag
    per (device, stream): text slices on side streams do not share state.  CPU tensors (the emulator build) are checked on the spot.
    `XCLIP_CHECK_TOKENS=1` checks synchronously on the GPU too (debugging)."""
    _per_stream = {}

    def __init__(self, device):
        self.flag = torch.zeros(1, dtype=torch.int32, device=device)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory() if device.type == "cuda" else None
        self.event = None
        self.launches = 0                 # embedding launches on this (device, stream) so far
        self.pending = 0                  # the launch whose flag copy `event` guards

    @classmethod
    def get(cls, device) -> "_TokenFlag":
        key = (device, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
        f = cls._per_stream.get(key)
        if f is None:
            f = cls._per_stream[key] = cls(device)
        return f

    def _raise(self, vocab):
        self.flag.zero_()
        if self.host is not None:
            self.host.zero_()
        self.event = None
        raise IndexError(f"index out of range in self: a text token id lies outside [0, {vocab}) (num_text_tokens) -- seen by embedding "
                         f"launch #{self.pending} on this stream (its output rows for those ids are NaN)")

    def poll(self, vocab, wait=False):
        """report what an EARLIER launch found, if its flag copy has already landed (wait=True: wait for it)"""
        if self.event is not None and (wait or self.event.query()):
            if wait:
                self.event.synchronize()
            self.event = None
            if int(self.host[0]) != 0:
                self._raise(vocab)

    def after_launch(self, vocab):
        import os
        self.launches += 1
        self.pending = self.launches
        self.vocab = vocab
        if self.host is None:
            if int(self.flag[0]) != 0:
                self._raise(vocab)
            return
        self.host.copy_(self.flag, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(self.flag.device))
        if os.environ.get("XCLIP_CHECK_TOKENS") == "1":
            self.poll(vocab, wait=True)


def check_tokens() -> None:
    """wait for the outstanding token-range checks of every stream and raise IndexError if an embedding launch saw an id outside its
    vocabulary (inference loops: call after a batch; training: the encoder's backward polls by itself)"""
    for tf in list(_TokenFlag._per_stream.values()):
        tf.poll(getattr(tf, "vocab", 0), wait=True)


def text_embed_fwd(tokens: Tensor, E: Tensor, P: Optional[Tensor], cls: Optional[Tensor]) -> Tensor:
    _dev_check(tokens, E, P, cls)
    assert tokens.dtype == torch.int64
    tokens = _c(tokens)
    b, n = tokens.shape
    vocab, dim = E.shape
    out = torch.empty(b, n + (1 if cls is not None else 0), dim, dtype=E.dtype, device=E.device)
    tf = _TokenFlag.get(E.device)
    tf.poll(vocab)
    _lib.check(_lib.lib().xclip_text_embed_fwd(tokens.data_ptr(), _c(E).data_ptr(), _ptr(None if P is None else _c(P)),
                                               _ptr(None if cls is None else _c(cls)), out.data_ptr(), b, n, dim, vocab,
                                               tf.flag.data_ptr(), dtype_code(E), _stream(E)), "xclip_text_embed_fwd")
    tf.after_launch(vocab)
    return out


def text_embed_bwd(dout: Tensor, tokens: Tensor, vocab: int, has_pos: bool, has_cls: bool, sorted_tokens=None):
    """-> fp32 accumulators dE [vocab, D], dP [n, D] | None, dcls [D] | None.
    sorted_tokens = (ids ascending [b*n] int64, perm [b*n] int64) (a torch.sort of tokens.flatten()): the embedding
    gradient is then a segmented sum over equal ids instead of one atomic per element."""
    _dev_check(dout, tokens)
    dout, tokens = _c(dout), _c(tokens)
    b, n = tokens.shape
    dim = dout.shape[-1]
    _TokenFlag.get(dout.device).poll(vocab)                  # the forward of this pass is long done: its range check has landed
    dE = torch.zeros(vocab, dim, dtype=torch.float32, device=dout.device)
    dP = torch.zeros(n, dim, dtype=torch.float32, device=dout.device) if has_pos else None
    dcls = torch.zeros(dim, dtype=torch.float32, device=dout.device) if has_cls else None
    L = _lib.lib()
    if has_pos or has_cls or sorted_tokens is None:
        _lib.check(L.xclip_text_embed_bwd(dout.data_ptr(), tokens.data_ptr(), 0 if sorted_tokens is not None else dE.data_ptr(),
                                          _ptr(dP), _ptr(dcls), b, n, dim, vocab, int(has_cls), dtype_code(dout), _stream(dout)),
                   "xclip_text_embed_bwd")
    if sorted_tokens is not None:
        ids, perm = sorted_tokens
        npos = n + int(has_cls)
        scatter_add_sorted(dout.view(b * npos, dim), ids, perm, dE, n_in=n, n_out=npos, row_off=int(has_cls))
    return dE, dP, dcls


def sort_ids(ids: Tensor, id_limit: int) -> Tuple[Tensor, Tensor]:
    """(ids ascending, perm) of a flat int64 id vector with values in [0, id_limit): perm[e] = where entry e stood; stable.  What
    `scatter_add_sorted` wants in front of it (the nn.Embedding backward of x_clip.py:320 as a segmented sum); sort.h's radix passes"""
    _dev_check(ids)
    assert ids.dtype == torch.int64 and ids.dim() == 1
    ids = _c(ids)
    n = ids.numel()
    out, perm = torch.empty_like(ids), torch.empty_like(ids)
    if n == 0:
        return out, perm
    L = _lib.lib()
    ws = workspace(ids.device, L.xclip_sort_ids_workspace_bytes(n) + 16)
    base = ws.data_ptr()
    off = (-base) % 16
    _lib.check(L.xclip_sort_ids(ids.data_ptr(), n, int(id_limit), out.data_ptr(), perm.data_ptr(), base + off, ws.numel() - off, _stream(ids)),
               "xclip_sort_ids")
    return out, perm


def scatter_add_sorted(src: Tensor, sorted_ids: Tensor, perm: Tensor, table: Tensor, n_in: int = 1, n_out: int = 1, row_off: int = 0):
    """table[sorted_ids[e]] += src[(perm[e] // n_in) * n_out + perm[e] % n_in + row_off]; sorted_ids ascending"""
    _dev_check(src, sorted_ids, perm, table)
    assert src.dim() == 2 and src.stride(1) == 1 and table.dtype == torch.float32 and table.is_contiguous()
    assert sorted_ids.dtype == torch.int64 and perm.dtype == torch.int64 and sorted_ids.is_contiguous() and perm.is_contiguous()
    assert sorted_ids.numel() == perm.numel() and table.shape[1] == src.shape[1]
    _lib.check(_lib.lib().xclip_scatter_add_sorted(src.data_ptr(), src.stride(0), sorted_ids.data_ptr(), perm.data_ptr(),
                                                   table.data_ptr(), table.shape[0], sorted_ids.numel(), src.shape[1], n_in, n_out, row_off,
                                                   dtype_code(src), _stream(src)), "xclip_scatter_add_sorted")


def patchify(image: Tensor, patch: int, keep: Optional[Tensor]) -> Tensor:
    """image [b, c, H, W] -> [b * nkeep, ceil16B(p*p*c)] rows ordered (p1 p2 c); keep: int32 [b, nkeep] or None"""
    _dev_check(image, keep)
    image = _c(image)
    b, c, H, W = image.shape
    npatch = (H // patch) * (W // patch)
    nkeep = npatch if keep is None else keep.shape[1]
    v = vec(image.dtype)
    ldo = (patch * patch * c + v - 1) // v * v
    out = torch.empty(b * nkeep, ldo, dtype=image.dtype, device=image.device)
    if keep is not None:
        assert keep.dtype == torch.int32
        keep = _c(keep)
    _lib.check(_lib.lib().xclip_patchify(image.data_ptr(), _ptr(keep), out.data_ptr(), ldo, b, c, H, W, patch, nkeep,
                                         dtype_code(image), _stream(image)), "xclip_patchify")
    return out


def token_mean_fwd(x: Tensor) -> Tensor:
    """x [b, n, D] (inner two dims contiguous, any batch stride) -> [b, D]"""
    _dev_check(x)
    b, n, dim = x.shape
    assert x.stride(2) == 1 and x.stride(1) == dim
    out = torch.empty(b, dim, dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().xclip_token_mean_fwd(x.data_ptr(), x.stride(0), out.data_ptr(), b, n, dim, dtype_code(x), _stream(x)),
               "xclip_token_mean_fwd")
    return out


def token_mean_bwd(dout: Tensor, n: int, add: Optional[Tensor] = None) -> Tensor:
    """dx[b, t] = dout[b] / n (+ add[b, t]); add: [b, n, D] with contiguous inner dims and any batch stride"""
    _dev_check(dout, add)
    dout = _c(dout)
    b, dim = dout.shape
    dx = torch.empty(b, n, dim, dtype=dout.dtype, device=dout.device)
    if add is not None:
        assert tuple(add.shape) == (b, n, dim) and add.stride(2) == 1 and add.stride(1) == dim and add.dtype == dout.dtype
    _lib.check(_lib.lib().xclip_token_mean_bwd(dout.data_ptr(), _ptr(add), 0 if add is None else add.stride(0), dx.data_ptr(), b, n,
                                               dim, dtype_code(dout), _stream(dout)), "xclip_token_mean_bwd")
    return dx


def copy_rows(src: Tensor, dst: Tensor) -> Tensor:
    """dst[r] = src[r] for 2-D views with unit inner stride and arbitrary row strides"""
    _dev_check(src, dst)
    assert src.dim() == 2 and dst.dim() == 2 and src.shape == dst.shape and src.stride(1) == 1 and dst.stride(1) == 1
    assert src.dtype == dst.dtype
    rows, dim = src.shape
    _lib.check(_lib.lib().xclip_copy_rows(src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), rows, dim, dtype_code(src),
                                          _stream(src)), "xclip_copy_rows")
    return dst


def add_rows(a: Tensor, b: Tensor) -> Tensor:
    _dev_check(a, b)
    a, b = _c(a), _c(b)
    assert a.shape == b.shape and a.dtype == b.dtype
    out = torch.empty_like(a)
    _lib.check(_lib.lib().xclip_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), dtype_code(a), _stream(a)), "xclip_add")
    return out


def rows_scatter_add(src: Tensor, idx: Optional[Tensor], table: Optional[Tensor], colsum: Optional[Tensor]):
    """table[idx[r]] += src[r] (fp32 table) and colsum += src.sum(0) (fp32); either accumulator may be None"""
    _dev_check(src, idx, table, colsum)
    assert src.dim() == 2 and src.stride(1) == 1
    rows, dim = src.shape
    if idx is not None:
        assert idx.dtype == torch.int32 and idx.numel() == rows and idx.is_contiguous()
    for t in (table, colsum):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    L = _lib.lib()
    ws = workspace(src.device, L.xclip_rows_scatter_add_workspace_bytes(rows, dim)) if colsum is not None else None
    _lib.check(L.xclip_rows_scatter_add(src.data_ptr(), src.stride(0), _ptr(idx), _ptr(table), _ptr(colsum), rows, dim, _ptr(ws),
                                        0 if ws is None else ws.numel(), dtype_code(src), _stream(src)), "xclip_rows_scatter_add")


def cast_from_f32(src: Tensor, dtype, scale: float = 1.0) -> Tensor:
    _dev_check(src)
    assert src.dtype == torch.float32 and src.is_contiguous()
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    _lib.check(_lib.lib().xclip_cast_from_f32(src.data_ptr(), dst.data_ptr(), src.numel(), scale, _DTYPES[dtype], _stream(src)),
               "xclip_cast_from_f32")
    return dst


# ---- GEMM -----------------------------------------------------------------------------------------------------------
# which production bf16 GEMM kernel the library carries: measurements that cannot be taken inside a run (the PMC traffic figure of
# profiles/gemm_traffic.json) are tagged with it and ignored by bench.py when they belong to an older kernel
GEMM_GENERATION = "gemm8a"   # (gemm8.h for the plain interior NT / NN products, gemm4.h for the rest; gemm9.h is a family of its own)

# True: forward / input-gradient products never get a split-K workspace, so the library cannot cut the row tail of a persistent launch off as
# a split-K problem (xclip_api.hip gemm2_tail_cut).  Which rows form that tail depends on the batch size; without the cut a sample's
# activations are BIT-identical whatever else shares its batch (tests/test_clip_gpu.py::test_full_size_properties_bf16), at the price of the
# last partial round of tiles (vision tower N = 512 products +12 ... +18 %, text +1 ... +2 %).  Weight gradients keep their split-K slabs.
BATCH_INVARIANT_GEMM = False


class KernelProbe:
    """Optional live measurement of the kernel families that make up a step (bench.py's `roofline` leg): while active, every
    xclip_gemm / attention / LayerNorm-family / contrastive-head (similarity forward and G) call is bracketed by HIP events on the stream it is launched on, and after every
    `clock_every`-th probed launch a one-wave clock sample (xclip_clock_sample, ~10 us) is queued on the same stream.
    `summary(family)` returns (launches, flops, seconds) with `algorithmic_bytes` set; `clock_mhz()` the sampled shader clocks."""
    active: Optional["KernelProbe"] = None

    def __init__(self, clock_every: int = 0, clock_slots: int = 512, under_load_every: int = 0, record: bool = True):
        self.records = []                 # (family, flops, ev0, ev1, algorithmic bytes, key)
        self.clock_every = clock_every
        self.clock_slots = clock_slots
        # under_load_every = n: in front of every n-th GEMM launch a one-wave sampler is started on a SIDE stream and counts shader cycles
        # for 200 us -- while the GEMM runs on the other 255 CUs (the sampler's CU cannot take a GEMM work-group meanwhile: such a pass
        # reads the clock the part sustains under the load, its kernel times are not to be quoted)
        self.under_load_every = under_load_every
        self.record = record
        self._clock = None
        self._clock_n = 0
        self._seen = 0
        self._side = None

    def __enter__(self):
        KernelProbe.active = self
        return self

    def __exit__(self, *exc):
        KernelProbe.active = None

    # -- recording (called by the wrappers below) --
    def _sample(self, t: Tensor, stream, ticks: int):
        if self._clock is None:
            self._clock = torch.zeros(self.clock_slots, 2, dtype=torch.int64, device=t.device)
        if self._clock_n < self.clock_slots:
            _lib.check(_lib.lib().xclip_clock_sample(self._clock[self._clock_n].data_ptr(), ticks, stream.cuda_stream), "xclip_clock_sample")
            self._clock_n += 1

    def begin(self, t: Tensor, family: str = ""):
        if self.under_load_every and family == "gemm":
            self._seen += 1
            if self._seen % self.under_load_every == 0:
                if self._side is None:
                    self._side = torch.cuda.Stream(device=t.device)
                    self._clock = torch.zeros(self.clock_slots, 2, dtype=torch.int64, device=t.device)
                # the sampler starts when the launching stream gets HERE (the host runs many kernels ahead of the device)
                self._side.wait_event(torch.cuda.current_stream(t.device).record_event())
                self._sample(t, self._side, 20000)
        if not self.record:
            return None
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(t.device))
        return ev0

    def end(self, t: Tensor, ev0, family: str, flops: float, nbytes: float, key=None):
        if not self.record:
            return
        ev1 = torch.cuda.Event(enable_timing=True)
        st = torch.cuda.current_stream(t.device)
        ev1.record(st)
        self.records.append((family, flops, ev0, ev1, nbytes, key))
        if self.clock_every and len(self.records) % self.clock_every == 0:
            self._sample(t, st, 1000)

    # -- results --
    def summary(self, family: str = "gemm"):
        torch.cuda.synchronize()
        recs = [r for r in self.records if r[0] == family]
        flops = sum(r[1] for r in recs)
        secs = sum(r[2].elapsed_time(r[3]) for r in recs) * 1e-3
        self.algorithmic_bytes = sum(r[4] for r in recs)
        return len(recs), flops, secs

    def by_shape(self):
        """{(M, N, K, layout, has_residual): [launches, milliseconds]} over the recorded GEMM launches"""
        torch.cuda.synchronize()
        out = {}
        for r in self.records:
            if r[0] != "gemm":
                continue
            e = out.setdefault(r[5], [0, 0.0])
            e[0] += 1
            e[1] += r[2].elapsed_time(r[3])
        return out

    def clock_mhz(self):
        """sorted shader-clock samples in MHz (cycles per 10 ns tick x 100); empty when no sample was taken"""
        if self._clock is None or self._clock_n == 0:
            return []
        torch.cuda.synchronize()
        c = self._clock[:self._clock_n].cpu()
        return sorted(float(cy) / max(float(tk), 1.0) * 100.0 for cy, tk in c.tolist() if tk > 0)


GemmProbe = KernelProbe            # (round 1-3 name)


def _probe(t: Tensor):
    p = KernelProbe.active
    return p if (p is not None and t.is_cuda) else None


def gemm(a: Tensor, b: Tensor, M: int, N: int, K: int, a_kmajor: bool = False, b_kmajor: bool = False, alpha: float = 1.0,
         bias: Optional[Tensor] = None, residual: Optional[Tensor] = None, addrows: Optional[Tensor] = None,
         rowidx: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """C[M, N] = alpha * op(a) op(b) (+ bias + addrows[rowidx] + residual); a, b 2-D with unit inner stride (views with a
    row stride are fine).  normal: a[M, K] / b[N, K];  k-major: a[K, M] / b[K, N]."""
    _dev_check(a, b, bias, residual, addrows, rowidx, out)
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1 and a.dtype == b.dtype
    assert tuple(a.shape) == ((K, M) if a_kmajor else (M, K)), (a.shape, M, K, a_kmajor)
    assert tuple(b.shape) == ((K, N) if b_kmajor else (N, K)), (b.shape, N, K, b_kmajor)
    if out is None:
        out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    assert out.dim() == 2 and out.stride(1) == 1 and tuple(out.shape) == (M, N)
    if residual is not None:
        assert residual.dim() == 2 and residual.stride(1) == 1 and tuple(residual.shape) == (M, N)
    if addrows is not None:
        assert rowidx is not None and rowidx.dtype == torch.int32 and rowidx.numel() == M and rowidx.is_contiguous()
        assert addrows.stride(1) == 1 and addrows.shape[1] == N
    L = _lib.lib()
    code = dtype_code(a)
    # split-K slabs: plain products, and the row tail of a long-K product with or without a skip term (xclip_api.hip gemm2_tail_cut)
    wbytes = L.xclip_gemm_workspace_bytes(M, N, K, code) if (bias is None and addrows is None and not (BATCH_INVARIANT_GEMM and not a_kmajor)) else 0
    ws = workspace(a.device, wbytes)
    probe = _probe(a)
    ev0 = probe.begin(a, "gemm") if probe is not None else None
    _lib.check(L.xclip_gemm(int(a_kmajor), int(b_kmajor), a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(),
                            out.stride(0), M, N, K, alpha, _ptr(bias), _ptr(residual),
                            0 if residual is None else residual.stride(0), _ptr(addrows), _ptr(rowidx),
                            0 if addrows is None else addrows.stride(0), _ptr(ws), 0 if ws is None else ws.numel(), code,
                            _stream(a)), "xclip_gemm")
    if probe is not None:
        esz = a.element_size()
        probe.end(a, ev0, "gemm", 2.0 * M * N * K, (M * K + N * K + M * N) * esz + (M * N * esz if residual is not None else 0),
                  (M, N, K, ("T" if a_kmajor else "N") + ("N" if b_kmajor else "T"), residual is not None))
    return out


def gemm_small_limit(max_flop: int = -1) -> int:
    """the FLOP bound (2 M N K) under which a bf16 product with a small output takes the latency-built 64 x 64 kernel (gemm_small.h);
    0 = never, < 0 = only ask.  -> the previous bound.  Process-wide (a tuning knob; the tests use it to reach the 256 x 256 kernels
    with small shapes)"""
    return int(_lib.lib().xclip_gemm_small_limit(int(max_flop)))


def bmm(a: Tensor, b: Tensor, M: int, N: int, K: int, a_kmajor: bool = False, b_kmajor: bool = False, alpha: float = 1.0) -> Tensor:
    """one launch of B independent products out[z] = alpha * op(a[z]) op(b[z]) -> [B, M, N], gemm()'s operand layouts per problem (normal
    a[z] [M, K] / b[z] [N, K]; k-major a[z] [K, M] / b[z] [K, N]); a, b 3-D with unit inner stride, N a multiple of the 16-byte chunk.  The
    fine-grained similarities of matched pairs and their gradients (reference inference return einsum('b t d, b i d -> b t i'), x_clip.py:742-743)"""
    _dev_check(a, b)
    assert a.dim() == 3 and b.dim() == 3 and a.shape[0] == b.shape[0] and a.dtype == b.dtype and a.stride(2) == 1 and b.stride(2) == 1
    assert N % vec(a.dtype) == 0
    B = a.shape[0]
    out = torch.empty(B, M, N, dtype=a.dtype, device=a.device)
    for z0 in range(0, B, 65535):
        z1 = min(B, z0 + 65535)
        _lib.check(_lib.lib().xclip_gemm_batched(int(a_kmajor), int(b_kmajor), a[z0:].data_ptr(), a.stride(1), a.stride(0), b[z0:].data_ptr(),
                                                 b.stride(1), b.stride(0), out[z0:].data_ptr(), N, M * N, z1 - z0, M, N, K, alpha,
                                                 dtype_code(a), _stream(a)), "xclip_gemm_batched")
    return out


def rowdot(a: Tensor, b: Tensor) -> Tensor:
    """out[r] = <a[r], b[r]> for a, b [rows, d] (einsum('b d, b d -> b'), x_clip.py:744-746)"""
    _dev_check(a, b)
    a, b = _c(a), _c(b)
    assert a.shape == b.shape and a.dim() == 2 and a.dtype == b.dtype
    out = torch.empty(a.shape[0], dtype=a.dtype, device=a.device)
    _lib.check(_lib.lib().xclip_rowdot(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), a.shape[0], a.shape[1],
                                       dtype_code(a), _stream(a)), "xclip_rowdot")
    return out


# ---- attention --------------------------------------------------------------------------------------------------------
def attention_fwd(qkv: Tensor, mask: Optional[Tensor], heads: int, scale: float, causal: bool = False,
                  head_dim: int = 64, dropout_p: float = 0.0, dropout_seed: int = 0) -> Tuple[Tensor, Tensor]:
    """qkv [b, n, 3*heads*head_dim] (head slots of 64 or 128 features); mask bool [b, n] or None; causal: key j hidden from query i < j
    -> out [b, n, heads*head_dim], lse fp32 [b, heads, n]"""
    _dev_check(qkv, mask)
    qkv = _c(qkv)
    b, n, w = qkv.shape
    assert head_dim in (64, 128) and w == 3 * heads * head_dim, "x_clip_amd attention kernels hold head slots of 64 or 128 features"
    out = torch.empty(b, n, heads * head_dim, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(b, heads, n, dtype=torch.float32, device=qkv.device)
    if mask is not None:
        assert mask.dtype == torch.bool and tuple(mask.shape) == (b, n)
        mask = _c(mask)
    probe = _probe(qkv)
    ev0 = probe.begin(qkv) if probe is not None else None
    _lib.check(_lib.lib().xclip_attention_fwd(qkv.data_ptr(), _ptr(mask), out.data_ptr(), lse.data_ptr(), b, n, heads, head_dim, scale,
                                              int(causal), float(dropout_p), int(dropout_seed) & _U64, dtype_code(qkv), _stream(qkv)),
               "xclip_attention_fwd")
    if probe is not None:      # S = Q K^T and O = P V: 2 x 2 n^2 hd per head (causal: the same count -- the skipped half is the kernel's gain); q, k, v in, o out
        probe.end(qkv, ev0, "attention", 4.0 * b * heads * n * n * head_dim, 4 * b * n * heads * head_dim * qkv.element_size() + 4 * b * heads * n, "attn_fwd")
    return out, lse


def attention_bwd(qkv: Tensor, mask: Optional[Tensor], out: Tensor, dout: Tensor, lse: Tensor, heads: int, scale: float,
                  causal: bool = False, head_dim: int = 64, dropout_p: float = 0.0, dropout_seed: int = 0) -> Tensor:
    _dev_check(qkv, mask, out, dout)
    dout = _c(dout)
    b, n, _ = qkv.shape
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(b, heads, n, dtype=torch.float32, device=qkv.device)
    probe = _probe(qkv)
    ev0 = probe.begin(qkv) if probe is not None else None
    _lib.check(_lib.lib().xclip_attention_bwd(qkv.data_ptr(), _ptr(mask), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                                              delta.data_ptr(), dqkv.data_ptr(), b, n, heads, head_dim, scale, int(causal), float(dropout_p),
                                              int(dropout_seed) & _U64, dtype_code(qkv), _stream(qkv)), "xclip_attention_bwd")
    if probe is not None:      # dV = P^T dO, dP = dO V^T, dQ = dS K, dK = dS^T Q (algorithmic: 4 products; the S recompute is the implementation's); q, k, v, o, dO in, dq, dk, dv out
        probe.end(qkv, ev0, "attention", 8.0 * b * heads * n * n * head_dim, 8 * b * n * heads * head_dim * qkv.element_size() + 8 * b * heads * n, "attn_bwd")
    return dqkv


def attention_pool_fwd(q: Tensor, kv: Tensor, mask: Optional[Tensor], heads: int, scale: float, head_dim: int = 64,
                       visible_keys: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """ONE query row per (sample, head): q [b, heads*head_dim], kv [b, n, 2*heads*head_dim] (keys | values of every row), mask bool [b, n] or None
    -> out [b, heads*head_dim], lse fp32 [b, heads].  visible_keys: keys [0, visible_keys) are seen (default n)"""
    _dev_check(q, kv, mask)
    q, kv = _c(q), _c(kv)
    b, n, w = kv.shape
    assert head_dim in (64, 128) and w == 2 * heads * head_dim and tuple(q.shape) == (b, heads * head_dim) and q.dtype == kv.dtype
    out = torch.empty_like(q)
    lse = torch.empty(b, heads, dtype=torch.float32, device=q.device)
    if mask is not None:
        assert mask.dtype == torch.bool and tuple(mask.shape) == (b, n)
        mask = _c(mask)
    probe = _probe(q)
    ev0 = probe.begin(q) if probe is not None else None
    _lib.check(_lib.lib().xclip_attention_pool_fwd(q.data_ptr(), kv.data_ptr(), _ptr(mask), out.data_ptr(), lse.data_ptr(), b, n, heads, head_dim,
                                                   scale, n if visible_keys is None else int(visible_keys), dtype_code(q), _stream(q)),
               "xclip_attention_pool_fwd")
    if probe is not None:      # one query: 4 n hd per head; k, v in (+ q), o out
        probe.end(q, ev0, "attention", 4.0 * b * heads * n * head_dim, (2 * b * n + 2 * b) * heads * head_dim * q.element_size(), "attn_pool_fwd")
    return out, lse


def attention_pool_bwd(q: Tensor, kv: Tensor, mask: Optional[Tensor], out: Tensor, dout: Tensor, lse: Tensor, heads: int, scale: float,
                       head_dim: int = 64, visible_keys: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """-> dq [b, heads*head_dim], dkv [b, n, 2*heads*head_dim] (every row written)"""
    _dev_check(q, kv, mask, out, dout, lse)
    q, kv, out, dout = _c(q), _c(kv), _c(out), _c(dout)
    b, n, _ = kv.shape
    assert lse.dtype == torch.float32 and lse.is_contiguous() and out.shape == q.shape and dout.shape == q.shape and dout.dtype == q.dtype
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    probe = _probe(q)
    ev0 = probe.begin(q) if probe is not None else None
    _lib.check(_lib.lib().xclip_attention_pool_bwd(q.data_ptr(), kv.data_ptr(), _ptr(None if mask is None else _c(mask)), out.data_ptr(),
                                                   dout.data_ptr(), lse.data_ptr(), dq.data_ptr(), dkv.data_ptr(), b, n, heads, head_dim, scale,
                                                   n if visible_keys is None else int(visible_keys), dtype_code(q), _stream(q)),
               "xclip_attention_pool_bwd")
    if probe is not None:      # dV = p dO, dP = dO v, dQ = dS k, dK = dS q: 8 n hd per head; k, v in, dk, dv out
        probe.end(q, ev0, "attention", 8.0 * b * heads * n * head_dim, (4 * b * n + 4 * b) * heads * head_dim * q.element_size(), "attn_pool_bwd")
    return dq, dkv


def dropout(x: Tensor, p: float, seed: int, out: Optional[Tensor] = None) -> Tensor:
    """y = x * keep / (1 - p) with the keep-mask of (seed, flat element index) (csrc/kernels/common.h drop_hash; reference nn.Dropout in
    FeedForward, x_clip.py:193-194).  The same call on the gradient is the backward.  out=x works in place."""
    _dev_check(x, out)
    assert x.is_contiguous()
    y = torch.empty_like(x) if out is None else out
    _lib.check(_lib.lib().xclip_dropout(x.data_ptr(), y.data_ptr(), x.numel(), float(p), int(seed) & _U64, dtype_code(x), _stream(x)), "xclip_dropout")
    return y


# ---- contrastive head ---------------------------------------------------------------------------------------------------
def _scale_args(scale, log_scale: Optional[Tensor]):
    if log_scale is not None:
        assert log_scale.dtype == torch.float32 and log_scale.numel() == 1
    return float(scale), _ptr(log_scale)


def simloss_fwd(q: Tensor, k: Tensor, scale: float, diag_off: int, dcl: bool, coef: float, loss_accum: Optional[Tensor],
                log_scale: Optional[Tensor] = None):
    """S = scale * exp(log_scale) * q k^T -> lse [nq] fp32, pos [nq] fp32;  loss_accum (fp32 scalar) += coef * sum(lse - pos)"""
    return simloss_chunked_fwd(q, [(k, 0)], scale, diag_off, dcl, coef, loss_accum, log_scale)


def simloss_chunked_fwd(q: Tensor, k_chunks, scale: float, diag_off: int, dcl: bool, coef: float, loss_accum: Optional[Tensor],
                        log_scale: Optional[Tensor] = None, before_chunk=None):
    """Same result as simloss_fwd(q, K) with K given as a list of (chunk [nk_c, d], first global column) in the order they
    should be consumed; `before_chunk(c)` (optional) runs before chunk c is launched, e.g. to wait for its all-gather."""
    _dev_check(q, *[kc for kc, _ in k_chunks], loss_accum, log_scale)
    q = _c(q)
    nq, d = q.shape
    L = _lib.lib()
    sc, lsp = _scale_args(scale, log_scale)
    slots = sum((kc.shape[0] + 63) // 64 for kc, _ in k_chunks)
    ws = workspace(q.device, 2 * slots * nq * 4)
    pos = torch.zeros(nq, dtype=torch.float32, device=q.device)
    lse = torch.empty(nq, dtype=torch.float32, device=q.device)
    slot0 = 0
    for c, (kc, col0) in enumerate(k_chunks):
        if before_chunk is not None:
            before_chunk(c)
        kc = _c(kc)
        nk = kc.shape[0]
        assert kc.shape[1] == d and kc.dtype == q.dtype
        probe = _probe(q)
        ev0 = probe.begin(q) if probe is not None else None
        _lib.check(L.xclip_simloss_partial(q.data_ptr(), kc.data_ptr(), nq, nk, d, sc, lsp, diag_off - col0, int(dcl), ws.data_ptr(),
                                           slot0, slots, pos.data_ptr(), dtype_code(q), _stream(q)), "xclip_simloss_partial")
        if probe is not None:      # S = q k^T once; both latent sets in, two fp32 partials per (row, 64-column slot) out
            probe.end(q, ev0, "head", 2.0 * nq * nk * d, (nq + nk) * d * q.element_size() + 8 * nq * ((nk + 63) // 64), "sim_fwd")
        slot0 += (nk + 63) // 64
    _lib.check(L.xclip_simloss_combine(ws.data_ptr(), nq, slots, pos.data_ptr(), lse.data_ptr(), _ptr(loss_accum), coef, _stream(q)),
               "xclip_simloss_combine")
    return lse, pos


def simloss_grad(q: Tensor, k: Tensor, scale: float, diag_off: int, dcl: bool, a: float, c: float, e: float, lse_q: Tensor,
                 lse_k: Tensor, dtau_accum: Optional[Tensor], log_scale: Optional[Tensor] = None, gmul: Optional[Tensor] = None,
                 times_scale: bool = False, out: Optional[Tensor] = None) -> Tensor:
    """-> G [nq, nk rounded up to the chunk] in q.dtype (padding columns are zero); dtau_accum += sum(G * S).
    `out`: a [nq, >= roundup(nk)] view (unit inner stride) to write into, e.g. a column block of a wider G."""
    _dev_check(q, k, lse_q, lse_k, dtau_accum, log_scale, gmul, out)
    q, k = _c(q), _c(k)
    nq, d = q.shape
    nk = k.shape[0]
    v = vec(q.dtype)
    ldg = (nk + v - 1) // v * v
    if out is None:
        G = torch.empty(nq, ldg, dtype=q.dtype, device=q.device)
    else:
        G = out
        assert G.dim() == 2 and G.stride(1) == 1 and G.shape[0] == nq and G.shape[1] >= ldg and G.dtype == q.dtype
    assert lse_q.dtype == torch.float32 and lse_k.dtype == torch.float32 and lse_q.numel() == nq and lse_k.numel() == nk
    assert lse_q.is_contiguous() and lse_k.is_contiguous()
    if gmul is not None:
        assert gmul.dtype == torch.float32 and gmul.numel() == 1
    sc, lsp = _scale_args(scale, log_scale)
    probe = _probe(q)
    ev0 = probe.begin(q) if probe is not None else None
    _lib.check(_lib.lib().xclip_simloss_grad(q.data_ptr(), k.data_ptr(), nq, nk, d, sc, lsp, diag_off, int(dcl), a, c, e, _ptr(gmul),
                                             int(times_scale), lse_q.data_ptr(), lse_k.data_ptr(), G.data_ptr(), G.stride(0),
                                             _ptr(dtau_accum), dtype_code(q), _stream(q)), "xclip_simloss_grad")
    if probe is not None:          # S recomputed once; both latent sets and the two lse vectors in, G out
        probe.end(q, ev0, "head", 2.0 * nq * nk * d, (nq + nk) * (d * q.element_size() + 4) + nq * ldg * q.element_size(), "sim_grad")
    return G


# ---- fine-grained (FILIP) head -----------------------------------------------------------------------------------------------
def filip_reduce(S: Tensor, mask: Tensor, log_temp: Tensor, t2i: Tensor, i2t: Tensor, kmax: Tensor, tmax: Tensor, cnt: Tensor,
                 nt: int, ni: int, yc: int, y0: int):
    """S [bx*nt, >= yc*ni] chunk of token similarities -> columns [y0, y0+yc) of t2i / i2t [bx, ytotal] (+ arg-max positions)"""
    _dev_check(S, mask, log_temp, t2i, i2t, kmax, tmax, cnt)
    bx = mask.shape[0]
    ytotal = t2i.shape[1]
    assert S.dim() == 2 and S.stride(1) == 1 and S.shape[0] == bx * nt and mask.dtype == torch.uint8 and mask.is_contiguous()
    assert kmax.dtype == torch.int16 and tuple(kmax.shape) == (bx, nt, ytotal) and tuple(tmax.shape) == (bx, ytotal, ni)
    assert t2i.dtype == torch.float32 and t2i.is_contiguous() and i2t.is_contiguous() and i2t.shape == t2i.shape
    _lib.check(_lib.lib().xclip_filip_reduce(S.data_ptr(), S.stride(0), mask.data_ptr(), log_temp.data_ptr(), t2i.data_ptr(),
                                             i2t.data_ptr(), ytotal, kmax.data_ptr(), tmax.data_ptr(), cnt.data_ptr(), bx, nt, yc, ni,
                                             y0, ytotal, dtype_code(S), _stream(S)), "xclip_filip_reduce")


def filip_fused_ok(nt: int, ni: int, d: int, dtype) -> bool:
    """can the token similarity + reductions run as ONE fused GEMM launch (filip5.h)?  bf16, d % 64 == 0, nt >= 32, ni >= 32"""
    return dtype in _DTYPES and bool(_lib.lib().xclip_filip_fused_ok(nt, ni, d, _DTYPES[dtype]))


def filip_fused_workspace_bytes(bx: int, nt: int, yc: int, ni: int) -> int:
    return int(_lib.lib().xclip_filip_fused_workspace_bytes(bx, nt, yc, ni))


def filip_fused_fwd(X: Tensor, mask: Tensor, Y: Tensor, log_temp: Tensor, t2i: Tensor, i2t: Tensor, kmax: Tensor, tmax: Tensor,
                    cnt: Tensor, ws: Tensor, y0: int):
    """X [bx, nt, d] text-token latents, Y [yc, ni, d] image-token latents of images [y0, y0 + yc) -> those columns of t2i / i2t
    [bx, ytotal] and of the arg-max maps, the token similarities never leaving the GEMM (x_clip.py:797-811)"""
    _dev_check(X, mask, Y, log_temp, t2i, i2t, kmax, tmax, cnt, ws)
    bx, nt, d = X.shape
    yc, ni, _ = Y.shape
    ytotal = t2i.shape[1]
    assert X.is_contiguous() and Y.is_contiguous() and X.dtype == Y.dtype and Y.shape[2] == d
    assert mask.dtype == torch.uint8 and mask.is_contiguous() and tuple(mask.shape) == (bx, nt)
    assert kmax.dtype == torch.int16 and tuple(kmax.shape) == (bx, nt, ytotal) and tuple(tmax.shape) == (bx, ytotal, ni)
    assert t2i.dtype == torch.float32 and t2i.is_contiguous() and i2t.is_contiguous() and i2t.shape == t2i.shape
    assert ws.dtype == torch.uint8 and ws.is_contiguous()
    _lib.check(_lib.lib().xclip_filip_fused_fwd(X.data_ptr(), mask.data_ptr(), Y.data_ptr(), log_temp.data_ptr(), t2i.data_ptr(), i2t.data_ptr(),
                                                ytotal, kmax.data_ptr(), tmax.data_ptr(), cnt.data_ptr(), ws.data_ptr(), ws.numel(), bx, nt, yc, ni,
                                                d, y0, ytotal, dtype_code(X), _stream(X)), "xclip_filip_fused_fwd")


def filip_route(P: Tensor, mask: Tensor, log_temp: Tensor, g1: Tensor, g2: Tensor, kmax: Tensor, tmax: Tensor, cnt: Tensor,
                nt: int, ni: int, yc: int, y0: int):
    """fills the chunk P [bx*nt, ldp] of d loss / d (token similarity) for image columns [y0, y0+yc)"""
    _dev_check(P, mask, log_temp, g1, g2, kmax, tmax, cnt)
    bx = mask.shape[0]
    ytotal = g1.shape[1]
    assert P.dim() == 2 and P.stride(1) == 1 and P.shape[0] == bx * nt and g1.is_contiguous() and g2.is_contiguous()
    _lib.check(_lib.lib().xclip_filip_route(P.data_ptr(), P.stride(0), mask.data_ptr(), log_temp.data_ptr(), g1.data_ptr(),
                                            g2.data_ptr(), ytotal, kmax.data_ptr(), tmax.data_ptr(), cnt.data_ptr(), bx, nt, yc, ni, y0,
                                            ytotal, dtype_code(P), _stream(P)), "xclip_filip_route")


def rowlse(S: Tensor, diag_off: int, dcl: bool, coef: float, loss_accum: Optional[Tensor]) -> Tensor:
    _dev_check(S, loss_accum)
    assert S.dim() == 2 and S.dtype == torch.float32 and S.stride(1) == 1
    rows, cols = S.shape
    lse = torch.empty(rows, dtype=torch.float32, device=S.device)
    _lib.check(_lib.lib().xclip_rowlse(S.data_ptr(), S.stride(0), rows, cols, diag_off, int(dcl), coef, lse.data_ptr(),
                                       _ptr(loss_accum), _stream(S)), "xclip_rowlse")
    return lse


def rowgrad(S: Tensor, lse: Tensor, diag_off: int, dcl: bool, coef: float, gmul: Optional[Tensor], dtau_accum: Optional[Tensor]) -> Tensor:
    _dev_check(S, lse, gmul, dtau_accum)
    rows, cols = S.shape
    G = torch.empty(rows, cols, dtype=torch.float32, device=S.device)
    _lib.check(_lib.lib().xclip_rowgrad(S.data_ptr(), S.stride(0), lse.data_ptr(), rows, cols, diag_off, int(dcl), coef, _ptr(gmul),
                                        G.data_ptr(), cols, _ptr(dtau_accum), _stream(S)), "xclip_rowgrad")
    return G


def simreg_diff(A: Tensor, C: Tensor, diag_off: int, sumsq_accum: Tensor) -> Tensor:
    """D = A - C with the global diagonal (column r + diag_off of row r) zeroed, written over A; *sumsq_accum += sum D^2.
    A, C [rows, cols] similarity blocks in the model dtype (x_clip.py:773-784)."""
    _dev_check(A, C, sumsq_accum)
    assert A.dim() == 2 and A.shape == C.shape and A.dtype == C.dtype and A.stride(1) == 1 and C.stride(1) == 1
    assert sumsq_accum.dtype == torch.float32
    rows, cols = A.shape
    _lib.check(_lib.lib().xclip_simreg_diff(A.data_ptr(), A.stride(0), C.data_ptr(), C.stride(0), A.data_ptr(), A.stride(0), rows, cols,
                                            diag_off, sumsq_accum.data_ptr(), dtype_code(A), _stream(A)), "xclip_simreg_diff")
    return A


def rotary_(x: Tensor, n: int, inv_freq: Tensor, inverse: bool = False, head_dim: int = 64) -> Tensor:
    """in-place rotary position embedding on x [rows, slots * head_dim] (packed q | k | v head slots of 64 or 128 features), position =
    row % n, angles pos * inv_freq[j] on the first rot = 2 * len(inv_freq) = min(dim_head, 32) features of every slot, pairs (j, j + rot / 2)
    (x_clip.py:155-176, 221-223, 311); inverse=True is the backward of the forward call"""
    _dev_check(x, inv_freq)
    assert x.dim() == 2 and x.stride(1) == 1 and head_dim in (64, 128) and x.shape[1] % head_dim == 0
    assert inv_freq.dtype == torch.float32 and 1 <= inv_freq.numel() <= 16 and inv_freq.is_contiguous()
    _lib.check(_lib.lib().xclip_rotary(x.data_ptr(), x.stride(0), x.shape[0], n, x.shape[1] // head_dim, head_dim, 2 * inv_freq.numel(),
                                       inv_freq.data_ptr(), int(inverse), dtype_code(x), _stream(x)), "xclip_rotary")
    return x


def dwconv4s2_fwd(x: Tensor, w: Tensor) -> Tensor:
    """depthwise 4x4 / stride 2 / pad 1 convolution over the token grid: x [b, h*h, C], w [C, 1, 4, 4] -> [b, (h/2)^2, C]
    (first stage of `downsample_image_embeds`, x_clip.py:560-568)"""
    _dev_check(x, w)
    b, n, C = x.shape
    h = math.isqrt(n)
    assert h * h == n and h % 2 == 0, "downsample_image_embeds needs an even-sided square token grid"
    assert x.is_contiguous() and w.is_contiguous() and w.numel() == C * 16 and w.dtype == x.dtype
    y = torch.empty(b, (h // 2) ** 2, C, dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().xclip_dwconv4s2_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), b, h, C, dtype_code(x), _stream(x)), "xclip_dwconv4s2_fwd")
    return y


def dwconv4s2_bwd(dy: Tensor, x: Tensor, w: Tensor, dw_accum: Tensor) -> Tensor:
    """-> dx [b, h*h, C]; dw_accum (fp32 [C * 16]) += the weight gradient"""
    _dev_check(dy, x, w, dw_accum)
    b, n, C = x.shape
    h = math.isqrt(n)
    assert dy.is_contiguous() and x.is_contiguous() and dw_accum.dtype == torch.float32 and dw_accum.numel() == C * 16
    L = _lib.lib()
    wbytes = L.xclip_dwconv4s2_workspace_bytes(b, h, C, dtype_code(x))
    ws = workspace(x.device, wbytes)
    dx = torch.empty_like(x)
    _lib.check(L.xclip_dwconv4s2_bwd(dy.data_ptr(), x.data_ptr(), w.data_ptr(), dx.data_ptr(), dw_accum.data_ptr(), ws.data_ptr(), wbytes,
                                     b, h, C, dtype_code(x), _stream(x)), "xclip_dwconv4s2_bwd")
    return dx


def gather_rows(src: Tensor, idx: Tensor) -> Tensor:
    """out[r] = src[idx[r]] for src [rows_in, D] (row stride free), idx int32 [rows]"""
    _dev_check(src, idx)
    assert src.dim() == 2 and src.stride(1) == 1 and idx.dtype == torch.int32 and idx.is_contiguous()
    out = torch.empty(idx.numel(), src.shape[1], dtype=src.dtype, device=src.device)
    _lib.check(_lib.lib().xclip_gather_rows(src.data_ptr(), src.stride(0), idx.data_ptr(), out.data_ptr(), idx.numel(), src.shape[1],
                                            dtype_code(src), _stream(src)), "xclip_gather_rows")
    return out


def cross_entropy_fwd(logits: Tensor, cols: int, labels: Tensor, loss_accum: Tensor) -> Tensor:
    """lse [rows] of logits[:, :cols]; *loss_accum += sum_r (lse[r] - logits[r, labels[r]])   (mlm.py:103-107)"""
    _dev_check(logits, labels, loss_accum)
    assert logits.dim() == 2 and logits.stride(1) == 1 and labels.dtype == torch.int64 and labels.is_contiguous()
    rows = logits.shape[0]
    lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
    _lib.check(_lib.lib().xclip_cross_entropy_fwd(logits.data_ptr(), logits.stride(0), labels.data_ptr(), rows, cols, lse.data_ptr(),
                                                  loss_accum.data_ptr(), dtype_code(logits), _stream(logits)), "xclip_cross_entropy_fwd")
    return lse


def cross_entropy_bwd_(logits: Tensor, cols: int, labels: Tensor, lse: Tensor, gmul: Tensor) -> Tensor:
    """in place: logits <- (gmul / rows) (softmax - onehot), padding columns zeroed"""
    _dev_check(logits, labels, lse, gmul)
    assert gmul.dtype == torch.float32 and gmul.numel() == 1
    _lib.check(_lib.lib().xclip_cross_entropy_bwd(logits.data_ptr(), logits.stride(0), labels.data_ptr(), lse.data_ptr(), gmul.data_ptr(),
                                                  logits.shape[0], cols, dtype_code(logits), _stream(logits)), "xclip_cross_entropy_bwd")
    return logits


def layernorm_chain_fwd(p: Tensor, g1: Tensor, res: Tensor, g2: Tensor):
    """x1 = LN(p) g1 + res, h2 = LN(x1) g2 in one pass (x_clip.py:245,288-289 + :126) -> (x1, mean1, rstd1, h2, mean2, rstd2)"""
    _dev_check(p, g1, res, g2)
    p, res = _c(p), _c(res)
    dim = p.shape[-1]
    rows = p.numel() // dim
    assert res.shape == p.shape and res.dtype == p.dtype and g1.numel() == dim and g2.numel() == dim
    x1, h2 = torch.empty_like(p), torch.empty_like(p)
    st = [torch.empty(rows, dtype=torch.float32, device=p.device) for _ in range(4)]
    probe = _probe(p)
    ev0 = probe.begin(p) if probe is not None else None
    _lib.check(_lib.lib().xclip_layernorm_chain_fwd(p.data_ptr(), _c(g1).data_ptr(), res.data_ptr(), x1.data_ptr(), st[0].data_ptr(),
                                                    st[1].data_ptr(), _c(g2).data_ptr(), h2.data_ptr(), st[2].data_ptr(), st[3].data_ptr(),
                                                    rows, dim, ln_eps(p.dtype), dtype_code(p), _stream(p)), "xclip_layernorm_chain_fwd")
    if probe is not None:      # reads p, res; writes x1, h2
        probe.end(p, ev0, "layernorm", 0.0, 4 * rows * dim * p.element_size(), "ln_chain_fwd")
    return x1, st[0], st[1], h2, st[2], st[3]


def layernorm_chain_bwd(dh2: Tensor, x1: Tensor, g2: Tensor, mean2: Tensor, rstd2: Tensor, dres: Tensor, p: Tensor, g1: Tensor,
                        mean1: Tensor, rstd1: Tensor, dg2: Tensor, dg1: Tensor):
    """-> (dx1 = LN2'(dh2) + dres, dp = LN1'(dx1)); dg2 / dg1 (fp32 [dim]) accumulate the gain gradients"""
    _dev_check(dh2, x1, g2, mean2, rstd2, dres, p, g1, mean1, rstd1, dg2, dg1)
    dh2, dres = _c(dh2), _c(dres)
    dim = x1.shape[-1]
    rows = x1.numel() // dim
    assert x1.is_contiguous() and p.is_contiguous() and dg2.dtype == torch.float32 and dg1.dtype == torch.float32
    dx1, dp = torch.empty_like(x1), torch.empty_like(x1)
    L = _lib.lib()
    wbytes = L.xclip_layernorm_chain_bwd_workspace_bytes(rows, dim)
    ws = workspace(x1.device, wbytes)
    probe = _probe(x1)
    ev0 = probe.begin(x1) if probe is not None else None
    _lib.check(L.xclip_layernorm_chain_bwd(dh2.data_ptr(), x1.data_ptr(), _c(g2).data_ptr(), mean2.data_ptr(), rstd2.data_ptr(),
                                           dres.data_ptr(), dx1.data_ptr(), p.data_ptr(), _c(g1).data_ptr(), mean1.data_ptr(),
                                           rstd1.data_ptr(), dp.data_ptr(), dg2.data_ptr(), dg1.data_ptr(), ws.data_ptr(), wbytes, rows, dim,
                                           dtype_code(x1), _stream(x1)), "xclip_layernorm_chain_bwd")
    if probe is not None:      # reads dh2, x1, dres, p; writes dx1, dp
        probe.end(x1, ev0, "layernorm", 0.0, 6 * rows * dim * x1.element_size(), "ln_chain_bwd")
    return dx1, dp


# ---- visual self-supervision head ---------------------------------------------------------------------------------------


def _f32(t: Optional[Tensor]) -> Optional[Tensor]:
    return None if t is None else _c(t.detach().float())


def batchnorm_fwd(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], running_mean: Optional[Tensor], running_var: Optional[Tensor],
                  momentum: float, eps: float, training: bool, relu: bool):
    """BatchNorm1d (+ ReLU) over the rows of x [rows, cols] -> (y, mean, rstd); running_mean / running_var (fp32, contiguous) are
    updated in place in training mode (visual_ssl.py:112-136)"""
    _dev_check(x, gamma, beta, running_mean, running_var)
    assert x.dim() == 2 and x.is_contiguous()
    for r in (running_mean, running_var):
        assert r is None or (r.dtype == torch.float32 and r.is_contiguous())
    rows, cols = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(cols, dtype=torch.float32, device=x.device)
    rstd = torch.empty(cols, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    wbytes = L.xclip_batchnorm_workspace_bytes(rows, cols)
    ws = workspace(x.device, wbytes)
    _lib.check(L.xclip_batchnorm_fwd(x.data_ptr(), _ptr(gamma), _ptr(beta), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(running_mean),
                                     _ptr(running_var), momentum, eps, int(training), int(relu), rows, cols, _ptr(ws), wbytes, dtype_code(x),
                                     _stream(x)), "xclip_batchnorm_fwd")
    return y, mean, rstd


def batchnorm_bwd(x: Tensor, dy: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], mean: Tensor, rstd: Tensor, training: bool, relu: bool,
                  need_affine_grads: bool):
    """-> (dx, dgamma, dbeta); the last two fp32 [cols] or None"""
    _dev_check(x, dy, gamma, beta, mean, rstd)
    dy = _c(dy)
    assert x.dim() == 2 and x.is_contiguous() and dy.shape == x.shape and dy.dtype == x.dtype
    rows, cols = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(cols, dtype=torch.float32, device=x.device) if need_affine_grads else None
    db = torch.empty(cols, dtype=torch.float32, device=x.device) if need_affine_grads else None
    L = _lib.lib()
    wbytes = L.xclip_batchnorm_workspace_bytes(rows, cols)
    ws = workspace(x.device, wbytes)
    _lib.check(L.xclip_batchnorm_bwd(x.data_ptr(), dy.data_ptr(), _ptr(gamma), _ptr(beta), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                     _ptr(dg), _ptr(db), int(training), int(relu), rows, cols, _ptr(ws), wbytes, dtype_code(x), _stream(x)),
               "xclip_batchnorm_bwd")
    return dx, dg, db


def neg_cosine_fwd(p: Tensor, z: Tensor, coef: float, loss_accum: Tensor):
    """*loss_accum += coef sum_r (2 - 2 cos(p_r, z_r)) -> (cos, 1/|p|, 1/|z|) per row (visual_ssl.py:104-107)"""
    _dev_check(p, z, loss_accum)
    assert p.dim() == 2 and p.is_contiguous() and z.is_contiguous() and z.shape == p.shape and z.dtype == p.dtype
    rows, dim = p.shape
    st = [torch.empty(rows, dtype=torch.float32, device=p.device) for _ in range(3)]
    _lib.check(_lib.lib().xclip_neg_cosine_fwd(p.data_ptr(), z.data_ptr(), rows, dim, coef, st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(),
                                               loss_accum.data_ptr(), dtype_code(p), _stream(p)), "xclip_neg_cosine_fwd")
    return st


def neg_cosine_bwd(p: Tensor, z: Tensor, st, gmul: Tensor, coef: float) -> Tensor:
    _dev_check(p, z, gmul)
    assert gmul.dtype == torch.float32 and gmul.numel() == 1
    dp = torch.empty_like(p)
    _lib.check(_lib.lib().xclip_neg_cosine_bwd(p.data_ptr(), z.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), gmul.data_ptr(),
                                               coef, dp.data_ptr(), p.shape[0], p.shape[1], dtype_code(p), _stream(p)), "xclip_neg_cosine_bwd")
    return dp


def ntxent_lse(q: Tensor, other: Tensor, scale: float, coef: float, loss_accum: Tensor) -> Tensor:
    """NT-Xent rows of `q` (nt_xent_loss, visual_ssl.py:90-102): logits scale * [q q^T with its diagonal removed | q other^T], the positive
    of row i is column i of the second block.  -> lse [nq] fp32;  *loss_accum += coef * sum_i (lse_i - scale <q_i, other_i>).
    Two xclip_simloss_partial passes into one slot table (the first excludes its diagonal and owns no positive), one combine."""
    _dev_check(q, other, loss_accum)
    q, other = _c(q), _c(other)
    nq, d = q.shape
    assert other.shape == q.shape and other.dtype == q.dtype
    L = _lib.lib()
    half = (nq + 63) // 64
    slots = 2 * half
    ws = workspace(q.device, 2 * slots * nq * 4)
    unused = torch.zeros(nq, dtype=torch.float32, device=q.device)
    pos = torch.zeros(nq, dtype=torch.float32, device=q.device)
    lse = torch.empty(nq, dtype=torch.float32, device=q.device)
    code, st = dtype_code(q), _stream(q)
    _lib.check(L.xclip_simloss_partial(q.data_ptr(), q.data_ptr(), nq, nq, d, scale, None, 0, 1, ws.data_ptr(), 0, slots, unused.data_ptr(), code, st),
               "xclip_simloss_partial")
    _lib.check(L.xclip_simloss_partial(q.data_ptr(), other.data_ptr(), nq, nq, d, scale, None, 0, 0, ws.data_ptr(), half, slots, pos.data_ptr(), code, st),
               "xclip_simloss_partial")
    _lib.check(L.xclip_simloss_combine(ws.data_ptr(), nq, slots, pos.data_ptr(), lse.data_ptr(), _ptr(loss_accum), coef, st), "xclip_simloss_combine")
    return lse
