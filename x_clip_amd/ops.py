"""Tensor-level wrappers over the C ABI: shape/dtype checks, output allocation (torch is the allocator and the
stream provider -- plumbing only), one library call each.  No arithmetic happens in Python."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

Tensor = torch.Tensor

_DTYPES = {torch.float32: 0, torch.bfloat16: 1}


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
    """grow-only scratch per device; kernels using it are ordered on the caller's stream"""
    if nbytes <= 0:
        return None
    ws = _workspaces.get(device)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _workspaces[device] = ws
    return ws


# ---- LayerNorm family ---------------------------------------------------------------------------------------------
def layernorm_fwd(x: Tensor, g: Tensor, res: Optional[Tensor] = None, geglu: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
    """x: [..., D] (or [..., 2D] with geglu) -> y [..., D], mean [rows], rstd [rows]"""
    _dev_check(x, g, res)
    x = _c(x)
    width = x.shape[-1]
    dim = width // 2 if geglu else width
    rows = x.numel() // width
    y = torch.empty(*x.shape[:-1], dim, dtype=x.dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    if res is not None:
        res = _c(res)
        assert res.shape == y.shape and res.dtype == x.dtype
    assert g.dtype == x.dtype and g.numel() == dim
    L = _lib.lib()
    _lib.check(L.xclip_layernorm_fwd(x.data_ptr(), width, _c(g).data_ptr(), _ptr(res), y.data_ptr(), mean.data_ptr(),
                                     rstd.data_ptr(), rows, dim, ln_eps(x.dtype), int(geglu), dtype_code(x), _stream(x)),
               "xclip_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy: Tensor, x: Tensor, g: Tensor, mean: Tensor, rstd: Tensor, geglu: bool = False) -> Tuple[Tensor, Tensor]:
    """-> dx (shape of x), dg (fp32 [D])"""
    _dev_check(dy, x, g)
    dy, x = _c(dy), _c(x)
    width = x.shape[-1]
    dim = width // 2 if geglu else width
    rows = x.numel() // width
    dx = torch.empty_like(x)
    dg = torch.zeros(dim, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    _lib.check(L.xclip_layernorm_bwd(dy.data_ptr(), x.data_ptr(), width, _c(g).data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                     dx.data_ptr(), width, dg.data_ptr(), rows, dim, int(geglu), dtype_code(x), _stream(x)),
               "xclip_layernorm_bwd")
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
def text_embed_fwd(tokens: Tensor, E: Tensor, P: Optional[Tensor], cls: Optional[Tensor]) -> Tensor:
    _dev_check(tokens, E, P, cls)
    assert tokens.dtype == torch.int64
    tokens = _c(tokens)
    b, n = tokens.shape
    dim = E.shape[1]
    out = torch.empty(b, n + (1 if cls is not None else 0), dim, dtype=E.dtype, device=E.device)
    _lib.check(_lib.lib().xclip_text_embed_fwd(tokens.data_ptr(), _c(E).data_ptr(), _ptr(None if P is None else _c(P)),
                                               _ptr(None if cls is None else _c(cls)), out.data_ptr(), b, n, dim,
                                               dtype_code(E), _stream(E)), "xclip_text_embed_fwd")
    return out


def text_embed_bwd(dout: Tensor, tokens: Tensor, vocab: int, has_pos: bool, has_cls: bool):
    """-> fp32 accumulators dE [vocab, D], dP [n, D] | None, dcls [D] | None"""
    _dev_check(dout, tokens)
    dout, tokens = _c(dout), _c(tokens)
    b, n = tokens.shape
    dim = dout.shape[-1]
    dE = torch.zeros(vocab, dim, dtype=torch.float32, device=dout.device)
    dP = torch.zeros(n, dim, dtype=torch.float32, device=dout.device) if has_pos else None
    dcls = torch.zeros(dim, dtype=torch.float32, device=dout.device) if has_cls else None
    _lib.check(_lib.lib().xclip_text_embed_bwd(dout.data_ptr(), tokens.data_ptr(), dE.data_ptr(), _ptr(dP), _ptr(dcls), b, n,
                                               dim, int(has_cls), dtype_code(dout), _stream(dout)), "xclip_text_embed_bwd")
    return dE, dP, dcls


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
    _dev_check(x)
    x = _c(x)
    b, n, dim = x.shape
    out = torch.empty(b, dim, dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().xclip_token_mean_fwd(x.data_ptr(), out.data_ptr(), b, n, dim, dtype_code(x), _stream(x)),
               "xclip_token_mean_fwd")
    return out


def token_mean_bwd(dout: Tensor, n: int, into: Optional[Tensor] = None) -> Tensor:
    _dev_check(dout, into)
    dout = _c(dout)
    b, dim = dout.shape
    dx = into if into is not None else torch.empty(b, n, dim, dtype=dout.dtype, device=dout.device)
    assert dx.is_contiguous()
    _lib.check(_lib.lib().xclip_token_mean_bwd(dout.data_ptr(), dx.data_ptr(), b, n, dim, int(into is not None),
                                               dtype_code(dout), _stream(dout)), "xclip_token_mean_bwd")
    return dx


def cast_from_f32(src: Tensor, dtype, scale: float = 1.0) -> Tensor:
    _dev_check(src)
    assert src.dtype == torch.float32 and src.is_contiguous()
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    _lib.check(_lib.lib().xclip_cast_from_f32(src.data_ptr(), dst.data_ptr(), src.numel(), scale, _DTYPES[dtype], _stream(src)),
               "xclip_cast_from_f32")
    return dst


# ---- GEMM -----------------------------------------------------------------------------------------------------------
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
    plain = bias is None and residual is None and addrows is None
    wbytes = L.xclip_gemm_workspace_bytes(M, N, K, code) if plain else 0
    ws = workspace(a.device, wbytes)
    _lib.check(L.xclip_gemm(int(a_kmajor), int(b_kmajor), a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(),
                            out.stride(0), M, N, K, alpha, _ptr(bias), _ptr(residual),
                            0 if residual is None else residual.stride(0), _ptr(addrows), _ptr(rowidx),
                            0 if addrows is None else addrows.stride(0), _ptr(ws), 0 if ws is None else ws.numel(), code,
                            _stream(a)), "xclip_gemm")
    return out


# ---- attention --------------------------------------------------------------------------------------------------------
def attention_fwd(qkv: Tensor, mask: Optional[Tensor], heads: int, scale: float) -> Tuple[Tensor, Tensor]:
    """qkv [b, n, 3*heads*64]; mask bool [b, n] or None -> out [b, n, heads*64], lse fp32 [b, heads, n]"""
    _dev_check(qkv, mask)
    qkv = _c(qkv)
    b, n, w = qkv.shape
    assert w == 3 * heads * 64, "x_clip_amd attention kernels are built for dim_head = 64"
    out = torch.empty(b, n, heads * 64, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(b, heads, n, dtype=torch.float32, device=qkv.device)
    if mask is not None:
        assert mask.dtype == torch.bool and tuple(mask.shape) == (b, n)
        mask = _c(mask)
    _lib.check(_lib.lib().xclip_attention_fwd(qkv.data_ptr(), _ptr(mask), out.data_ptr(), lse.data_ptr(), b, n, heads, scale,
                                              dtype_code(qkv), _stream(qkv)), "xclip_attention_fwd")
    return out, lse


def attention_bwd(qkv: Tensor, mask: Optional[Tensor], out: Tensor, dout: Tensor, lse: Tensor, heads: int, scale: float) -> Tensor:
    _dev_check(qkv, mask, out, dout)
    dout = _c(dout)
    b, n, _ = qkv.shape
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(b, heads, n, dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.lib().xclip_attention_bwd(qkv.data_ptr(), _ptr(mask), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                                              delta.data_ptr(), dqkv.data_ptr(), b, n, heads, scale, dtype_code(qkv),
                                              _stream(qkv)), "xclip_attention_bwd")
    return dqkv


# ---- contrastive head ---------------------------------------------------------------------------------------------------
def simloss_fwd(q: Tensor, k: Tensor, scale: float, diag_off: int, dcl: bool, coef: float, loss_accum: Optional[Tensor]):
    """-> lse [nq] fp32, pos [nq] fp32;  loss_accum (fp32 scalar tensor) += coef * sum(lse - pos)"""
    _dev_check(q, k, loss_accum)
    q, k = _c(q), _c(k)
    nq, d = q.shape
    nk = k.shape[0]
    L = _lib.lib()
    ws = workspace(q.device, L.xclip_simloss_workspace_bytes(nq, nk))
    pos = torch.zeros(nq, dtype=torch.float32, device=q.device)
    lse = torch.empty(nq, dtype=torch.float32, device=q.device)
    _lib.check(L.xclip_simloss_fwd(q.data_ptr(), k.data_ptr(), nq, nk, d, scale, diag_off, int(dcl), coef, ws.data_ptr(),
                                   pos.data_ptr(), lse.data_ptr(), _ptr(loss_accum), dtype_code(q), _stream(q)),
               "xclip_simloss_fwd")
    return lse, pos


def simloss_grad(q: Tensor, k: Tensor, scale: float, diag_off: int, dcl: bool, a: float, c: float, e: float, lse_q: Tensor,
                 lse_k: Tensor, dtau_accum: Tensor) -> Tensor:
    """-> G [nq, nk rounded up to the chunk] in q.dtype (padding columns are zero); dtau_accum += sum(G * S)"""
    _dev_check(q, k, lse_q, lse_k, dtau_accum)
    q, k = _c(q), _c(k)
    nq, d = q.shape
    nk = k.shape[0]
    v = vec(q.dtype)
    ldg = (nk + v - 1) // v * v
    G = torch.empty(nq, ldg, dtype=q.dtype, device=q.device)
    _lib.check(_lib.lib().xclip_simloss_grad(q.data_ptr(), k.data_ptr(), nq, nk, d, scale, diag_off, int(dcl), a, c, e,
                                             lse_q.data_ptr(), lse_k.data_ptr(), G.data_ptr(), ldg, dtau_accum.data_ptr(),
                                             dtype_code(q), _stream(q)), "xclip_simloss_grad")
    return G
