"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package (x_clip_amd).

A CPU restatement of the lucidrains/x-clip contrastive-training hot path
(`CLIP.forward(text, image, return_loss=True)`, reference x_clip/x_clip.py:597-875) written as pure
functions over a flat `state_dict` (reference key names).  Arithmetic is plain
torch on the CPU in whatever dtype the state dict / inputs carry (fp32 or fp64); gradients come from
torch autograd over these functions, plus an independent closed-form numpy statement of the
similarity/InfoNCE/DCL head in `simloss_closed_form`.

Parity pin: `tests/test_oracle_golden.py` checks every function here against golden vectors that
`oracle/make_golden.py` produced by importing and running the *reference itself* in the build container
(committed under tests/golden/).  Only tests/, `__graft_entry__.smoke()` and bench.py's cpu_baseline
leg may import this module.

Each function cites the reference lines it restates.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, asdict
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------------
# configuration = the reference constructor's keyword arguments (x_clip.py:413-456) that reach the
# hot path.  Defaults are the reference's defaults.
# --------------------------------------------------------------------------------------------------
@dataclass
class ClipConfig:
    dim_text: int = 512
    dim_image: int = 512
    dim_latent: int = 512
    num_text_tokens: int = 10000
    text_enc_depth: int = 6
    text_seq_len: int = 256
    text_heads: int = 8
    text_dim_head: int = 64
    text_has_cls_token: bool = True
    text_pad_id: int = 0
    text_rotary_pos_emb: bool = False
    text_causal_mask: bool = False              # autoregressive text encoder: no CLS token, causal attention, EOS pooling
    text_eos_id: Optional[int] = None
    visual_enc_depth: int = 6
    visual_heads: int = 8
    visual_dim_head: int = 64
    visual_image_size: int = 256
    visual_patch_size: int = 32
    visual_has_cls_token: bool = True
    channels: int = 3
    use_all_token_embeds: bool = False
    downsample_image_embeds: bool = False
    decoupled_contrastive_learning: bool = False
    extra_latent_projection: bool = False
    multiview_loss_weight: float = 0.1
    sim_reg_loss_weight: float = 0.0
    use_mlm: bool = False
    text_ssl_loss_weight: float = 0.05
    use_visual_ssl: bool = False                # SimSiam side loss (visual_ssl.py:207-259) through CLIP(visual_ssl = module)
    image_ssl_loss_weight: float = 0.05
    visual_ssl_type: str = "simsiam"            # "simclr": NT-Xent between two views (visual_ssl.py:263-299), projector hidden width 4096
    simclr_temperature: float = 0.1
    ssl_projection_size: int = 256              # SimSiam(projection_size, projection_hidden_size): not CLIP keywords (the CLIP
    ssl_projection_hidden_size: int = 4096      # constructors swallow them in **kwargs)

    @property
    def num_patches(self) -> int:
        return (self.visual_image_size // self.visual_patch_size) ** 2

    def ctor_kwargs(self) -> dict:
        """kwargs for the reference / product constructor (patch dropout is passed separately)."""
        return asdict(self)

    def vit_kwargs(self, patch_dropout: float = 0.0) -> dict:
        """VisionTransformer(**...) as CLIP builds it (x_clip.py:496-507): the visual-SSL cases construct the encoder first and
        hand it to both SimSiam(net) and CLIP(image_encoder =)"""
        return dict(dim=self.dim_image, image_size=self.visual_image_size, patch_size=self.visual_patch_size,
                    channels=self.channels, depth=self.visual_enc_depth, heads=self.visual_heads,
                    dim_head=self.visual_dim_head, patch_dropout=patch_dropout)


# --------------------------------------------------------------------------------------------------
# row ops
# --------------------------------------------------------------------------------------------------
_LN_EPS_OVERRIDE = None


class layer_norm_eps:
    """`with layer_norm_eps(1e-3):` -- evaluate the oracle in fp64 AS THE MODEL OF A bf16 RUN: the reference's LayerNorm picks its
    epsilon from the STORAGE dtype (1e-3 for anything that is not fp32, x_clip.py:118), so the fp64 yardstick of a bf16 product
    run has to use the bf16 epsilon, otherwise the comparison carries a model difference inside its tolerance."""

    def __init__(self, eps: float):
        self.eps = eps

    def __enter__(self):
        global _LN_EPS_OVERRIDE
        self.prev, _LN_EPS_OVERRIDE = _LN_EPS_OVERRIDE, self.eps
        return self

    def __exit__(self, *exc):
        global _LN_EPS_OVERRIDE
        _LN_EPS_OVERRIDE = self.prev
        return False


def layer_norm(x: Tensor, g: Tensor) -> Tensor:
    """Gain-only LayerNorm with a dtype dependent epsilon (x_clip.py:112-121):
    biased variance, eps 1e-5 for fp32 (we also use it for fp64), 1e-3 for every other dtype."""
    eps = 1e-5 if x.dtype in (torch.float32, torch.float64) else 1e-3
    if _LN_EPS_OVERRIDE is not None:
        eps = _LN_EPS_OVERRIDE
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * g


def geglu(y: Tensor) -> Tensor:
    """GEGLU (x_clip.py:180-183): first half of the last dim is the value, second half the gate;
    exact (erf) GELU on the gate."""
    half = y.shape[-1] // 2
    val, gate = y[..., :half], y[..., half:]
    return val * (0.5 * gate * (1.0 + torch.erf(gate / math.sqrt(2.0))))


def l2_normalize(x: Tensor) -> Tensor:
    """F.normalize(dim=-1) (x_clip.py:54-55): x / max(||x||, 1e-12)."""
    n = torch.sqrt((x * x).sum(dim=-1, keepdim=True))
    return x / n.clamp_min(1e-12)


# --------------------------------------------------------------------------------------------------
# transformer block stack (x_clip.py:201-291)
# --------------------------------------------------------------------------------------------------
def rotary_freqs(seq_len: int, dim_head: int, dtype=torch.float32) -> Tensor:
    """RotaryEmbedding(min(dim_head, 32)).forward(seq_len) (x_clip.py:155-166): [seq_len, rot] angles, the rot/2
    frequencies repeated twice."""
    rot = min(dim_head, 32)
    inv_freq = 1.0 / (10000 ** (torch.arange(0, rot, 2).float() / rot))
    f = torch.einsum("i,j->ij", torch.arange(seq_len).float(), inv_freq)
    return torch.cat((f, f), dim=-1).to(dtype)


def apply_rotary(freqs: Tensor, t: Tensor) -> Tensor:
    """apply_rotary_pos_emb (x_clip.py:168-176): the first rot features are rotated pairwise (j, j + rot/2), the rest pass."""
    rot = freqs.shape[-1]
    a, rest = t[..., :rot], t[..., rot:]
    x1, x2 = a[..., : rot // 2], a[..., rot // 2:]
    half = torch.cat((-x2, x1), dim=-1)
    return torch.cat((a * freqs.cos() + half * freqs.sin(), rest), dim=-1)


def dropout_keep(seed: int, count: int, p: float) -> Tensor:
    """the keep-mask the product's kernels use for element indices 0 .. count-1 under `seed` (x_clip_amd/csrc/kernels/common.h
    drop_hash: a stateless 32-bit mix of the seed words and the 64-bit element index; keep iff hash >= p * 2^32), rebuilt with numpy.
    This is NOT the reference's RNG stream (torch's Philox cannot be matched from a fused kernel): the parity tests evaluate the
    reference's dropout arithmetic (x_clip.py:193-194,241: mask / (1 - p) after the softmax / after the inner LayerNorm) on THIS mask."""
    M = np.uint64(0xFFFFFFFF)
    idx = np.arange(count, dtype=np.uint64)
    seed = np.uint64(seed & ((1 << 64) - 1))
    with np.errstate(over="ignore"):
        h = (seed & M) ^ (((seed >> np.uint64(32)) * np.uint64(0x85EBCA77)) & M) ^ (((idx & M) * np.uint64(0x9E3779B1)) & M) \
            ^ (((idx >> np.uint64(32)) * np.uint64(0xC2B2AE3D)) & M)
        h ^= h >> np.uint64(16); h = (h * np.uint64(0x7FEB352D)) & M
        h ^= h >> np.uint64(15); h = (h * np.uint64(0x846CA68B)) & M
        h ^= h >> np.uint64(16)
    thresh = 0 if p <= 0 else int(np.float32(p).astype(np.float64) * 4294967296.0)
    return torch.from_numpy((h >= np.uint64(thresh)))


def attention(x: Tensor, sd: Dict[str, Tensor], pfx: str, heads: int, dim_head: int,
              key_mask: Optional[Tensor], rotary: Optional[Tensor] = None, causal: bool = False,
              drop: Optional[Tuple[float, int]] = None) -> Tensor:
    """Attention.forward (x_clip.py:213-245): bias-free fused qkv projection, q scaled by
    dim_head**-0.5, key padding mask, optional causal mask (:231-234), softmax in fp32 (or wider), bias-free out
    projection followed by a LayerNorm."""
    b, n, _ = x.shape
    qkv = x @ sd[pfx + "to_qkv.weight"].t()                     # [b, n, 3*h*d]
    qkv = qkv.view(b, n, 3, heads, dim_head).permute(2, 0, 3, 1, 4)   # [3, b, h, n, d]
    q, k, v = qkv[0] * (dim_head ** -0.5), qkv[1], qkv[2]
    if rotary is not None:                                      # x_clip.py:221-223: q, k AND v
        q, k, v = (apply_rotary(rotary.to(x.dtype), t) for t in (q, k, v))
    scores = q @ k.transpose(-1, -2)                            # [b, h, n, n]
    if key_mask is not None:
        scores = scores.masked_fill(~key_mask[:, None, None, :], -torch.finfo(scores.dtype).max)
    if causal:
        scores = scores.masked_fill(torch.ones(n, n, dtype=torch.bool).triu(1), -torch.finfo(scores.dtype).max)
    sm_dtype = torch.float32 if scores.dtype != torch.float64 else torch.float64
    probs = torch.softmax(scores.to(sm_dtype), dim=-1).to(scores.dtype)
    if drop is not None and drop[0] > 0:                        # Attention.dropout (x_clip.py:212,241), mask over (b, h, i, j)
        keep = dropout_keep(drop[1], b * heads * n * n, drop[0]).view(b, heads, n, n)
        probs = probs * keep.to(probs.dtype) / (1.0 - float(np.float32(drop[0])))
    o = (probs @ v).permute(0, 2, 1, 3).reshape(b, n, heads * dim_head)
    o = o @ sd[pfx + "to_out.0.weight"].t()
    return layer_norm(o, sd[pfx + "to_out.1.g"])


def feed_forward(x: Tensor, sd: Dict[str, Tensor], pfx: str, drop: Optional[Tuple[float, int]] = None) -> Tensor:
    """FeedForward (x_clip.py:185-199): Linear(D, 8D) -> GEGLU -> LayerNorm(4D) -> Dropout -> Linear(4D, D), no biases."""
    y = x @ sd[pfx + "net.0.weight"].t()
    h = layer_norm(geglu(y), sd[pfx + "net.2.g"])
    if drop is not None and drop[0] > 0:                        # net.3 (x_clip.py:194), mask over the flat [rows, 4D] activation
        keep = dropout_keep(drop[1], h.numel(), drop[0]).view(h.shape)
        h = h * keep.to(h.dtype) / (1.0 - float(np.float32(drop[0])))
    return h @ sd[pfx + "net.4.weight"].t()


def transformer(x: Tensor, sd: Dict[str, Tensor], pfx: str, depth: int, heads: int, dim_head: int,
                key_mask: Optional[Tensor], rotary: Optional[Tensor] = None, causal: bool = False,
                dropout: Optional[Tuple[float, float, int]] = None) -> Tensor:
    """Transformer.forward (x_clip.py:274-291): norm_in, pre-norm residual attention + feed-forward
    blocks, norm_out."""
    x = layer_norm(x, sd[pfx + "norm_in.g"])
    for l in range(depth):
        a = f"{pfx}layers.{l}.0."
        f = f"{pfx}layers.{l}.1."
        da = (dropout[0], dropout[2] + 2 * l) if dropout is not None else None          # (attn p, ff p, the pass's seed): layer l uses
        df = (dropout[1], dropout[2] + 2 * l + 1) if dropout is not None else None      # seed + 2 l / seed + 2 l + 1, as the product does
        x = attention(layer_norm(x, sd[a + "norm.g"]), sd, a + "fn.", heads, dim_head, key_mask, rotary, causal, da) + x
        x = feed_forward(layer_norm(x, sd[f + "norm.g"]), sd, f + "fn.", df) + x
    return layer_norm(x, sd[pfx + "norm_out.g"])


# --------------------------------------------------------------------------------------------------
# encoders
# --------------------------------------------------------------------------------------------------
def encode_text(sd: Dict[str, Tensor], cfg: ClipConfig, tokens: Tensor, mask: Optional[Tensor]) -> Tensor:
    """TextTransformer.forward (x_clip.py:317-338): token embedding + absolute positions, learned CLS
    prepended (its mask slot is True), then the block stack.  Returns [b, n+1, dim_text]."""
    pfx = "text_transformer."
    b, n = tokens.shape
    x = sd[pfx + "token_emb.weight"][tokens]
    rotary = None
    if cfg.text_rotary_pos_emb:                                 # x_clip.py:311-312,328-330: no absolute table, n + 1 positions
        rotary = rotary_freqs(n + 1, cfg.text_dim_head)
    else:
        x = x + sd[pfx + "abs_pos_emb.weight"][:n][None]
    if not cfg.text_causal_mask:                                # the causal encoder has no CLS token (x_clip.py:314,332-337)
        cls = sd[pfx + "cls_token"].expand(b, 1, -1)
        x = torch.cat([cls, x], dim=1)
        if mask is not None:
            mask = torch.cat([torch.ones(b, 1, dtype=torch.bool), mask], dim=1)
    return transformer(x, sd, pfx + "transformer.", cfg.text_enc_depth, cfg.text_heads,
                       cfg.text_dim_head, mask, rotary, cfg.text_causal_mask)


def eos_to_front(enc: Tensor, tokens: Tensor, eos_id: int) -> Tensor:
    """CLIP.forward's post-processing of the causal encoder (x_clip.py:670-685; the `b` those lines use is the batch size): the
    encoding at each row's FIRST eos token moves to position 0, the remaining positions keep their order."""
    b, n, _ = enc.shape
    first = (tokens == eos_id).float().argmax(dim=-1)
    rows = []
    for i in range(b):
        e = int(first[i])
        rows.append(torch.cat([enc[i, e: e + 1], enc[i, :e], enc[i, e + 1:]], dim=0))
    return torch.stack(rows)


def patchify(image: Tensor, p: int) -> Tensor:
    """'b c (h p1) (w p2) -> b (h w) (p1 p2 c)' (x_clip.py:357)."""
    b, c, H, W = image.shape
    x = image.view(b, c, H // p, p, W // p, p)           # b c h p1 w p2
    x = x.permute(0, 2, 4, 3, 5, 1)                      # b h w p1 p2 c
    return x.reshape(b, (H // p) * (W // p), p * p * c)


def encode_image(sd: Dict[str, Tensor], cfg: ClipConfig, image: Tensor,
                 keep_idx: Optional[Tensor] = None) -> Tensor:
    """VisionTransformer.forward (x_clip.py:372-390): patch embedding (Linear with bias) + position
    table, optional patch dropout expressed as an explicit kept-index set `keep_idx` [b, n_keep]
    (x_clip.py:140-151 draws it from randn().topk; the oracle takes it as an input so both sides can
    share it), block stack, CLS = Linear(mean over tokens) prepended.  Returns [b, 1+n_keep, dim]."""
    pfx = "visual_transformer."
    x = patchify(image, cfg.visual_patch_size) @ sd[pfx + "to_tokens.1.weight"].t() \
        + sd[pfx + "to_tokens.1.bias"]
    x = x + sd[pfx + "pos_emb.weight"][None, : x.shape[1]]
    if keep_idx is not None:
        x = torch.gather(x, 1, keep_idx[..., None].expand(-1, -1, x.shape[-1]))
    out = transformer(x, sd, pfx + "transformer.", cfg.visual_enc_depth, cfg.visual_heads,
                      cfg.visual_dim_head, None)
    cls = out.mean(dim=1) @ sd[pfx + "to_cls_tokens.1.weight"].t()
    return torch.cat([cls[:, None], out], dim=1)


# --------------------------------------------------------------------------------------------------
# contrastive head
# --------------------------------------------------------------------------------------------------
def _info_nce(t2i: Tensor, i2t: Tensor, dcl: bool) -> Tensor:
    """InfoNCE / DCL tail (x_clip.py:821-847) for stacked [P, B, B] logits.  No max subtraction, the
    `+1e-20` inside the logs is kept."""
    B = t2i.shape[-1]
    e1, e2 = t2i.exp(), i2t.exp()
    pos1 = torch.diagonal(e1, dim1=-2, dim2=-1)
    pos2 = torch.diagonal(e2, dim1=-2, dim2=-1)
    if dcl:
        off = ~torch.eye(B, dtype=torch.bool)
        e1, e2 = e1 * off, e2 * off
    l1 = (-(pos1 + 1e-20).log() + (e1.sum(-1) + 1e-20).log()).mean(-1)
    l2 = (-(pos2 + 1e-20).log() + (e2.sum(-1) + 1e-20).log()).mean(-1)
    return (l1 + l2) / 2


def contrastive_loss(cfg: ClipConfig, temperature: Tensor,
                     t_lat: Tensor, i_lat: Tensor,
                     t_lat_x: Optional[Tensor], i_lat_x: Optional[Tensor],
                     text_mask: Optional[Tensor], m: int, n: int, ssl_weight: float = 0.0) -> Tensor:
    """x_clip.py:736,750-755,797-868.  `t_lat` is [(m b), d] (CLS mode) or [(m b), nt, d] (FILIP);
    `i_lat` likewise with n views.  `*_x` are the CLOOB extra-projection latents or None."""
    temp = temperature.exp()
    tl = t_lat.view(m, -1, *t_lat.shape[1:])
    il = i_lat.view(n, -1, *i_lat.shape[1:])
    tlx = tl if t_lat_x is None else t_lat_x.view(m, -1, *t_lat_x.shape[1:])
    ilx = il if i_lat_x is None else i_lat_x.view(n, -1, *i_lat_x.shape[1:])
    if cfg.use_all_token_embeds:
        # fine-grained (FILIP) similarity, x_clip.py:799-811
        s1 = torch.einsum("mxtd,nyid->mnxyti", tl, il) * temp
        s2 = s1 if t_lat_x is None else torch.einsum("mxtd,nyid->mnxyti", tlx, ilx) * temp
        tm = text_mask.view(m, -1, text_mask.shape[-1])                  # [m, b, t]
        w = tm[:, None, :, None, :]                                      # m 1 b 1 t
        t2i_tok = s1.max(dim=-1).values                                  # m n x y t
        t2i = (t2i_tok * w).sum(-1) / w.sum(-1).clamp(min=1e-6)
        masked = s2.masked_fill(~w[..., None], -torch.finfo(s2.dtype).max)
        i2t = masked.max(dim=-2).values.mean(dim=-1)                     # m n x y
    else:
        t2i = torch.einsum("mtd,nid->mnti", tl, il) * temp
        if t_lat_x is None:
            i2t = t2i.transpose(-1, -2)
        else:
            i2t = torch.einsum("mtd,nid->mnit", tlx, ilx) * temp
    B = t2i.shape[-1]
    losses = _info_nce(t2i.reshape(m * n, B, B), i2t.reshape(m * n, B, B),
                       cfg.decoupled_contrastive_learning)
    multiview = (m > 1) or (n > 1)
    w_mv = cfg.multiview_loss_weight if multiview else 0.0
    loss = losses[0] * (1.0 - (ssl_weight + w_mv))                        # cl_loss_weight, x_clip.py:855
    if multiview:
        loss = loss + losses[1:].mean() * w_mv
    if cfg.sim_reg_loss_weight > 0:
        loss = loss + sim_reg_loss(tl, il, tlx, ilx) * cfg.sim_reg_loss_weight     # x_clip.py:872-873
    return loss


def sim_reg_loss(tl: Tensor, il: Tensor, tlx: Tensor, ilx: Tensor) -> Tensor:
    """Similarity regularisation (x_clip.py:773-784): mean squared difference between the off-diagonal text-text and
    image-image similarities, for the main and the CLOOB-extra latents, averaged.  The reference's boolean mask has a
    leading dimension of 1, so it is only defined for a single view (m = n = 1), CLS-mode latents [1, B, d], and -- because
    its `*_extra` tensors are only reshaped under `extra_latent_projection` -- only with the extra projections on."""
    assert tl.shape[0] == 1 and il.shape[0] == 1 and tl.dim() == 3, "sim-reg loss: single view, CLS-mode latents"
    B = tl.shape[1]
    off = ~torch.eye(B, dtype=torch.bool)

    def sim(t):
        return (t[0] @ t[0].t())[off]

    mse = torch.nn.functional.mse_loss
    return (mse(sim(tl), sim(il)) + mse(sim(tlx), sim(ilx))) / 2


def downsample_latents(tokens: Tensor, w_dw: Tensor, w_pw: Tensor, b_pw: Tensor) -> Tensor:
    """`downsample_image_embeds` projection (x_clip.py:560-568): the image tokens [b, n, C] (n a perfect square) are laid out as
    a sqrt(n) x sqrt(n) grid, filtered per channel by a 4 x 4 stride-2 pad-1 convolution (weight [C, 1, 4, 4], no bias), then mapped
    to the latent width by a 1 x 1 convolution (weight [L, C, 1, 1] + bias): [b, n / 4, L]."""
    b, n, C = tokens.shape
    h = int(math.isqrt(n))
    assert h * h == n, "downsample_image_embeds needs a square token grid"
    x = tokens.transpose(1, 2).reshape(b, C, h, h)
    x = torch.nn.functional.conv2d(x, w_dw, None, stride=2, padding=1, groups=C)
    x = torch.nn.functional.conv2d(x, w_pw, b_pw)
    return x.flatten(2).transpose(1, 2)


def mlm_loss(sd: Dict[str, Tensor], cfg: ClipConfig, masked_seq: Tensor, labels: Tensor, text_mask: Tensor) -> Tensor:
    """MLM.forward after the random masking (mlm.py:96-109): the SAME text transformer encodes the masked sequence, `to_logits`
    (Linear with bias, vocabulary = num_text_tokens) scores every non-CLS position, cross-entropy averaged over the positions
    whose label is not the pad id.  The masking itself (mlm.py:70-94) is random; callers pass its outcome."""
    emb = encode_text(sd, cfg, masked_seq, text_mask)
    logits = emb[:, 1:] @ sd["mlm.to_logits.weight"].t() + sd["mlm.to_logits.bias"]
    return torch.nn.functional.cross_entropy(logits.transpose(1, 2), labels, ignore_index=cfg.text_pad_id)


def ssl_aug_one(x: Tensor) -> Tensor:
    """deterministic stand-ins for the two random augmentations of SimSiam (augment_fn / augment_fn2, visual_ssl.py:216-226):
    the fixtures, the oracle and the product tests all pass these two callables"""
    return x.flip(-1)


def ssl_aug_two(x: Tensor) -> Tensor:
    return 0.8 * x + 0.2 * x.roll(3, dims=-2)


SSL_PROJECTOR = "visual_ssl.online_encoder.projector."
SSL_PREDICTOR = "visual_ssl.online_predictor."
SIMCLR_PROJECTOR = "visual_ssl.net.projector."


class SslAugPair:
    """SimCLR takes ONE augmentation callable and calls it once per view (visual_ssl.py:291-295): this one alternates between the two
    deterministic stand-ins (odd calls: ssl_aug_one, even calls: ssl_aug_two)"""

    def __init__(self):
        self.calls = 0

    def __call__(self, x):
        self.calls += 1
        return ssl_aug_one(x) if self.calls % 2 == 1 else ssl_aug_two(x)


def ssl_projector(sd: Dict[str, Tensor], cfg: ClipConfig, P: str, img: Tensor, stats: Dict[str, list]) -> Tensor:
    """NetWrapper.forward with hidden_layer = -1 (visual_ssl.py:197-203): every token row of the encoder output through SimSiamMLP
    (:122-135: Linear - BN - ReLU - Linear - BN - ReLU - Linear - BN(affine = False), no biases)"""
    rep = encode_image(sd, cfg, img)
    x = rep.reshape(-1, rep.shape[-1])
    z1 = batch_norm_rows(x @ sd[P + "0.weight"].t(), sd[P + "1.weight"], sd[P + "1.bias"], stats[P + "1"])
    z2 = batch_norm_rows(torch.relu(z1) @ sd[P + "3.weight"].t(), sd[P + "4.weight"], sd[P + "4.bias"], stats[P + "4"])
    if "relu_margin" in stats:                                  # smallest |pre-activation|: how close any unit sits to the ReLU kink
        stats["relu_margin"].append(min(float(z1.detach().abs().min()), float(z2.detach().abs().min())))
    return batch_norm_rows(torch.relu(z2) @ sd[P + "6.weight"].t(), None, None, stats[P + "7"])


def _running_after(sd, stats, running):
    for k, seq in stats.items():
        rm, rv = sd[k + ".running_mean"].detach().clone(), sd[k + ".running_var"].detach().clone()
        for mean, var in seq:
            rm = 0.9 * rm + 0.1 * mean
            rv = 0.9 * rv + 0.1 * var
        running[k + ".running_mean"], running[k + ".running_var"] = rm, rv


def nt_xent(queries: Tensor, keys: Tensor, temperature: float) -> Tensor:
    """nt_xent_loss (visual_ssl.py:90-102): P = [queries; keys] (n = 2b rows), logits P P^T with the diagonal REMOVED, divided by the
    temperature; row i's label is its partner (i + b, shifted by the removed diagonal entry; i - b for the second half); summed
    cross-entropy / n.  The projections are not normalised."""
    b = queries.shape[0]
    n = 2 * b
    projs = torch.cat((queries, keys))
    logits = projs @ projs.t()
    logits = logits[~torch.eye(n, dtype=torch.bool)].reshape(n, n - 1) / temperature
    labels = torch.cat((torch.arange(b) + b - 1, torch.arange(b)))
    return torch.nn.functional.cross_entropy(logits, labels, reduction="sum") / n


def simclr_loss(sd: Dict[str, Tensor], cfg: ClipConfig, image: Tensor, running: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """SimCLR.forward (visual_ssl.py:289-299) with hidden_layer = -1: both views through the same NetWrapper, NT-Xent over the
    flattened projections (every token row is a sample)"""
    P = SIMCLR_PROJECTOR
    stats: Dict[str, list] = {P + "1": [], P + "4": [], P + "7": []}
    if running is not None:
        stats["relu_margin"] = []
    queries = ssl_projector(sd, cfg, P, ssl_aug_one(image), stats)
    keys = ssl_projector(sd, cfg, P, ssl_aug_two(image), stats)
    if running is not None:
        running["relu_margin"] = min(stats.pop("relu_margin"))
        _running_after(sd, stats, running)
    return nt_xent(queries, keys, cfg.simclr_temperature)


def batch_norm_rows(x: Tensor, g: Optional[Tensor], b: Optional[Tensor], stats: Optional[list] = None) -> Tensor:
    """nn.BatchNorm1d in train() mode over the rows of x [R, C] (biased variance, eps 1e-5); `stats` collects (mean, unbiased
    variance) -- what the running statistics are updated with"""
    mean = x.mean(dim=0)
    var = x.var(dim=0, unbiased=False)
    if stats is not None:
        stats.append((mean.detach(), x.var(dim=0, unbiased=True).detach()))
    y = (x - mean) * torch.rsqrt(var + 1e-5)
    if g is not None:
        y = y * g + b
    return y


def simsiam_loss(sd: Dict[str, Tensor], cfg: ClipConfig, image: Tensor, running: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """SimSiam.forward (visual_ssl.py:237-259) around the CLIP vision tower with hidden_layer = -1 (the CLIP default, x_clip.py:551):
    the representation is the encoder output [b, 1 + n, dim], every token row is projected (NetWrapper.forward :197-203,
    SimSiamMLP :122-135: Linear - BN - ReLU - Linear - BN - ReLU - Linear - BN(affine = False)), the predictor is MLP :112-120
    (Linear - BN - ReLU - Linear, with biases), loss_fn :104-107 is 2 - 2 cos, the targets are the projections themselves under
    no_grad (the target encoder IS the online encoder, :243-249; equal values because the encoder is deterministic without patch
    dropout).  `running`: if given, receives the BatchNorm running statistics after the step (four passes through the projector
    -- view one, view two, and both again for the targets -- and two through the predictor, momentum 0.1)."""
    P, Q = SSL_PROJECTOR, SSL_PREDICTOR
    stats: Dict[str, list] = {P + "1": [], P + "4": [], P + "7": [], Q + "1": []}

    def project(img):
        return ssl_projector(sd, cfg, P, img, stats)

    def predict(x):
        x = torch.relu(batch_norm_rows(x @ sd[Q + "0.weight"].t() + sd[Q + "0.bias"], sd[Q + "1.weight"], sd[Q + "1.bias"], stats[Q + "1"]))
        return x @ sd[Q + "3.weight"].t() + sd[Q + "3.bias"]

    one, two = ssl_aug_one(image), ssl_aug_two(image)
    proj_one, proj_two = project(one), project(two)
    pred_one, pred_two = predict(proj_one), predict(proj_two)
    for k in (P + "1", P + "4", P + "7"):                                  # the two target passes see the same batches again
        stats[k] = stats[k] + stats[k]
    cos = torch.nn.functional.cosine_similarity
    loss = (2 - 2 * cos(pred_one, proj_two.detach(), dim=-1, eps=1e-12)) + (2 - 2 * cos(pred_two, proj_one.detach(), dim=-1, eps=1e-12))
    if running is not None:
        _running_after(sd, stats, running)
    return loss.mean()


def clip_forward(sd: Dict[str, Tensor], cfg: ClipConfig, text: Tensor, image: Tensor,
                 aug_text: Sequence[Tensor] = (), aug_image: Sequence[Tensor] = (),
                 keep_idx: Optional[Tensor] = None, return_latents: bool = False,
                 mlm_masked: Optional[Tuple[Tensor, Tensor]] = None, ssl_running: Optional[Dict[str, Tensor]] = None):
    """CLIP.forward(return_loss=True) (x_clip.py:597-875) with its two side losses: MLM (`mlm_masked` = the masked sequence and
    labels the random masking produced, x_clip.py:620-622) and SimSiam (x_clip.py:623, deterministic augmentations)."""
    text_ssl = image_ssl = None
    if cfg.use_mlm and not return_latents:
        text_ssl = mlm_loss(sd, cfg, mlm_masked[0], mlm_masked[1], text != cfg.text_pad_id)
    if cfg.use_visual_ssl and not return_latents:
        image_ssl = (simclr_loss if cfg.visual_ssl_type == "simclr" else simsiam_loss)(sd, cfg, image, ssl_running)
    m, n = 1 + len(aug_text), 1 + len(aug_image)
    text = torch.cat([text, *aug_text], dim=0)
    image = torch.cat([image, *aug_image], dim=0)
    text_mask = text != cfg.text_pad_id                                   # x_clip.py:614
    enc_t = encode_text(sd, cfg, text, text_mask)
    if cfg.text_causal_mask:
        enc_t = eos_to_front(enc_t, text, cfg.text_eos_id)
    enc_i = encode_image(sd, cfg, image, keep_idx)
    if cfg.use_all_token_embeds:                                          # x_clip.py:702-709
        et = enc_t[:, 1:] if cfg.text_has_cls_token else enc_t
        ei = enc_i[:, 1:] if cfg.visual_has_cls_token else enc_i
    else:
        et, ei = enc_t[:, 0], enc_i[:, 0]
    def visual_latent(pfx):
        if cfg.downsample_image_embeds:                                   # x_clip.py:560-568
            return downsample_latents(ei, sd[pfx + ".1.weight"], sd[pfx + ".2.weight"], sd[pfx + ".2.bias"])
        return ei @ sd[pfx + ".weight"].t()

    tl = l2_normalize(et @ sd["to_text_latent.weight"].t())               # x_clip.py:713-715
    il = l2_normalize(visual_latent("to_visual_latent"))
    tlx = ilx = None
    if cfg.extra_latent_projection:                                       # x_clip.py:720-724
        tlx = l2_normalize(et @ sd["to_text_latent_extra.weight"].t())
        ilx = l2_normalize(visual_latent("to_visual_latent_extra"))
    if return_latents:
        return (tl, il) if tlx is None else (tl, il, tlx, ilx)
    loss = contrastive_loss(cfg, sd["temperature"], tl, il, tlx, ilx, text_mask, m, n,
                            ssl_weight=(cfg.text_ssl_loss_weight if cfg.use_mlm else 0.0)
                            + (cfg.image_ssl_loss_weight if cfg.use_visual_ssl else 0.0))
    if text_ssl is not None:
        loss = loss + text_ssl * cfg.text_ssl_loss_weight                 # x_clip.py:857-860
    if image_ssl is not None:
        loss = loss + image_ssl * cfg.image_ssl_loss_weight
    return loss


# --------------------------------------------------------------------------------------------------
# closed form of the CLS-mode head in numpy/fp64, independent of autograd
# --------------------------------------------------------------------------------------------------
def simloss_closed_form(T: np.ndarray, I: np.ndarray, tau: float, dcl: bool,
                        Tx: Optional[np.ndarray] = None, Ix: Optional[np.ndarray] = None):
    """Loss and gradients of  L = 1/2 [ mean_i(lse_j S_ij - S_ii) + mean_j(lse_i S'_ij - S'_jj) ]
    with S = e^tau T I^T and S' = S (or e^tau Tx Ix^T with the CLOOB extra latents); DCL drops the diagonal
    from both log-sum-exps (x_clip.py:813-847).  Returns dict(loss, dT, dI, dTx, dIx, dtau, lse_row,
    lse_col) in fp64."""
    T = np.asarray(T, np.float64); I = np.asarray(I, np.float64)
    B = T.shape[0]
    temp = math.exp(tau)
    S1 = temp * T @ I.T
    extra = Tx is not None
    S2 = temp * np.asarray(Tx, np.float64) @ np.asarray(Ix, np.float64).T if extra else S1
    eye = np.eye(B, dtype=bool)

    def lse(S, axis):
        E = np.exp(S)
        if dcl:
            E = np.where(eye, 0.0, E)
        return np.log(E.sum(axis=axis))

    lse_row = lse(S1, 1)           # over images j for each text i
    lse_col = lse(S2, 0)           # over texts i for each image j
    loss = 0.5 * ((lse_row - np.diag(S1)).mean() + (lse_col - np.diag(S2)).mean())
    off = (~eye) if dcl else np.ones_like(eye)
    G1 = np.exp(S1 - lse_row[:, None]) * off / (2 * B) - eye / (2 * B)     # dL/dS1
    G2 = np.exp(S2 - lse_col[None, :]) * off / (2 * B) - eye / (2 * B)     # dL/dS2
    out = dict(loss=loss, lse_row=lse_row, lse_col=lse_col)
    if extra:
        Tx = np.asarray(Tx, np.float64); Ix = np.asarray(Ix, np.float64)
        out.update(dT=temp * G1 @ I, dI=temp * G1.T @ T, dTx=temp * G2 @ Ix, dIx=temp * G2.T @ Tx,
                   dtau=(G1 * S1).sum() + (G2 * S2).sum())
    else:
        G = G1 + G2
        out.update(dT=temp * G @ I, dI=temp * G.T @ T, dTx=None, dIx=None, dtau=(G * S1).sum())
    return out


# --------------------------------------------------------------------------------------------------
# deterministic, platform independent parameter / input generation shared by the fixture generator
# and the tests (numpy legacy RandomState streams are frozen across numpy versions)
# --------------------------------------------------------------------------------------------------
def _ssl_aliases(cfg: ClipConfig):
    """state_dict prefixes under which the SSL module lists the (shared) vision tower again"""
    return ("visual_ssl.net.net.",) if cfg.visual_ssl_type == "simclr" else ("visual_ssl.net.", "visual_ssl.online_encoder.net.")


def state_dict_shapes(cfg: ClipConfig) -> Dict[str, Tuple[int, ...]]:
    """Key -> shape map of the default-built reference model."""
    shapes: Dict[str, Tuple[int, ...]] = {"temperature": ()}

    def tower(pfx, dim, depth, heads, dim_head):
        inner = heads * dim_head
        for l in range(depth):
            a, f = f"{pfx}layers.{l}.0.", f"{pfx}layers.{l}.1."
            shapes[a + "norm.g"] = (dim,)
            shapes[a + "fn.to_qkv.weight"] = (3 * inner, dim)
            shapes[a + "fn.to_out.0.weight"] = (dim, inner)
            shapes[a + "fn.to_out.1.g"] = (dim,)
            shapes[f + "norm.g"] = (dim,)
            shapes[f + "fn.net.0.weight"] = (8 * dim, dim)
            shapes[f + "fn.net.2.g"] = (4 * dim,)
            shapes[f + "fn.net.4.weight"] = (dim, 4 * dim)
        shapes[pfx + "norm_in.g"] = (dim,)
        shapes[pfx + "norm_out.g"] = (dim,)

    t = "text_transformer."
    if not cfg.text_causal_mask:
        shapes[t + "cls_token"] = (cfg.dim_text,)
    shapes[t + "token_emb.weight"] = (cfg.num_text_tokens + (1 if cfg.use_mlm else 0), cfg.dim_text)     # x_clip.py:487
    if cfg.text_rotary_pos_emb:                                 # buffer of RotaryEmbedding (x_clip.py:158-159)
        shapes[t + "rotary_pos_emb.inv_freq"] = (min(cfg.text_dim_head, 32) // 2,)
    else:
        shapes[t + "abs_pos_emb.weight"] = (cfg.text_seq_len, cfg.dim_text)
    tower(t + "transformer.", cfg.dim_text, cfg.text_enc_depth, cfg.text_heads, cfg.text_dim_head)
    v = "visual_transformer."
    pd = cfg.channels * cfg.visual_patch_size ** 2
    shapes[v + "to_tokens.1.weight"] = (cfg.dim_image, pd)
    shapes[v + "to_tokens.1.bias"] = (cfg.dim_image,)
    shapes[v + "pos_emb.weight"] = (cfg.num_patches, cfg.dim_image)
    tower(v + "transformer.", cfg.dim_image, cfg.visual_enc_depth, cfg.visual_heads, cfg.visual_dim_head)
    shapes[v + "to_cls_tokens.1.weight"] = (cfg.dim_image, cfg.dim_image)
    if cfg.use_mlm:                                             # MLM head (mlm.py:65); its `transformer` IS text_transformer: the
        shapes["mlm.to_logits.weight"] = (cfg.num_text_tokens, cfg.dim_text)      # state_dict lists those tensors a second time
        shapes["mlm.to_logits.bias"] = (cfg.num_text_tokens,)
        for k in [k for k in shapes if k.startswith(t)]:
            shapes["mlm.transformer." + k[len(t):]] = shapes[k]
    if cfg.use_visual_ssl:                                      # SimSiam (visual_ssl.py:122-135 projector, :112-120 predictor); `net` and
        H, ps = cfg.ssl_projection_hidden_size, cfg.ssl_projection_size   # `online_encoder.net` list the vision tower again
        P, Q = SSL_PROJECTOR, SSL_PREDICTOR

        def bn(pfx, width, affine=True):
            if affine:
                shapes[pfx + ".weight"] = (width,)
                shapes[pfx + ".bias"] = (width,)
            shapes[pfx + ".running_mean"] = (width,)
            shapes[pfx + ".running_var"] = (width,)
            shapes[pfx + ".num_batches_tracked"] = ()

        simclr = cfg.visual_ssl_type == "simclr"
        if simclr:                                              # SimCLR(net, project_dim): NetWrapper's default hidden width (visual_ssl.py:142,280)
            P, H = SIMCLR_PROJECTOR, 4096
        shapes[P + "0.weight"] = (H, cfg.dim_image)
        bn(P + "1", H)
        shapes[P + "3.weight"] = (H, H)
        bn(P + "4", H)
        shapes[P + "6.weight"] = (ps, H)
        bn(P + "7", ps, affine=False)
        if not simclr:
            shapes[Q + "0.weight"] = (H, ps)
            shapes[Q + "0.bias"] = (H,)
            bn(Q + "1", H)
            shapes[Q + "3.weight"] = (ps, H)
            shapes[Q + "3.bias"] = (ps,)
        for k in [k for k in shapes if k.startswith(v)]:
            for alias in _ssl_aliases(cfg):
                shapes[alias + k[len(v):]] = shapes[k]
    for k, d in (("to_text_latent", cfg.dim_text), ("to_visual_latent", cfg.dim_image)):
        for sfx in ("", "_extra"):
            if k == "to_visual_latent" and cfg.downsample_image_embeds:   # Sequential(RearrangeImage, Conv2d dw, Conv2d 1x1, Rearrange)
                shapes[k + sfx + ".1.weight"] = (d, 1, 4, 4)
                shapes[k + sfx + ".2.weight"] = (cfg.dim_latent, d, 1, 1)
                shapes[k + sfx + ".2.bias"] = (cfg.dim_latent,)
            else:
                shapes[k + sfx + ".weight"] = (cfg.dim_latent, d)
    return shapes


def make_state_dict(cfg: ClipConfig, seed: int, dtype=torch.float32) -> Dict[str, Tensor]:
    """Seeded synthetic parameters with reference-like scales: linear weights U(-1/sqrt(fan_in), +),
    embeddings / cls N(0, 1), LayerNorm gains 1 + 0.1 N(0, 1) (so gain gradients are exercised),
    temperature 1.0.  Keys are generated in sorted order from one RandomState stream."""
    rs = np.random.RandomState(seed)
    sd: Dict[str, Tensor] = {}
    for key, shape in sorted(state_dict_shapes(cfg).items()):
        if key.startswith("mlm.transformer.") or (cfg.use_visual_ssl and key.startswith(_ssl_aliases(cfg))):
            continue                                            # aliases of text_transformer.* / visual_transformer.*, filled in below
        if key.endswith("num_batches_tracked"):
            sd[key] = torch.tensor(0, dtype=torch.long)
            continue
        if key.startswith("visual_ssl.") and len(shape) == 1 and not key.endswith(".0.bias") and not key.endswith(".3.bias"):
            if key.endswith("running_var"):                     # BatchNorm1d tensors: gains ~1, shifts / running means small
                a = 1.0 + 0.2 * np.abs(rs.standard_normal(shape))
            elif key.endswith(".weight"):
                a = 1.0 + 0.1 * rs.standard_normal(shape)
            else:
                a = 0.1 * rs.standard_normal(shape)
        elif key == "temperature":
            a = np.array(1.0)
        elif key.endswith("inv_freq"):
            rot = 2 * shape[0]
            a = (1.0 / (10000 ** (torch.arange(0, rot, 2).float() / rot))).double().numpy()
        elif key.endswith(".g"):
            a = 1.0 + 0.1 * rs.standard_normal(shape)
        elif key.endswith(".bias"):
            a = rs.uniform(-0.05, 0.05, shape)
        elif "emb" in key or key.endswith("cls_token"):
            a = rs.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) == 4 else shape[-1]     # conv weights: in/groups * kh * kw
            bound = 1.0 / math.sqrt(fan_in)
            a = rs.uniform(-bound, bound, shape)
        sd[key] = torch.tensor(a, dtype=torch.float64).to(dtype)
    if cfg.use_mlm:
        for k in [k for k in sd if k.startswith("text_transformer.")]:
            sd["mlm.transformer." + k[len("text_transformer."):]] = sd[k]
    if cfg.use_visual_ssl:
        for k in [k for k in sd if k.startswith("visual_transformer.")]:
            for alias in _ssl_aliases(cfg):
                sd[alias + k[len("visual_transformer."):]] = sd[k]
    return sd


def make_inputs(cfg: ClipConfig, batch: int, seed: int, n_aug_text: int = 0, n_aug_image: int = 0,
                pad_tail: int = 3):
    """Seeded synthetic batch: randint tokens (a few trailing pad ids so the key mask is live) and
    N(0,1) images (fp64, cast by the caller)."""
    rs = np.random.RandomState(seed)

    def toks():
        t = rs.randint(1, cfg.num_text_tokens, size=(batch, cfg.text_seq_len))
        for r in range(batch):
            k = (r * 7 + 1) % (pad_tail + 1)
            if k:
                t[r, -k:] = cfg.text_pad_id
            if cfg.text_causal_mask:                            # every row ends in the eos id (before its padding); row 1 has an
                t[r, t.shape[1] - k - 1] = cfg.text_eos_id      # earlier one as well: the FIRST eos is the pooled position
                if r == 1:
                    t[r, 5] = cfg.text_eos_id
        return torch.tensor(t, dtype=torch.int64)

    def img():
        return torch.tensor(rs.standard_normal((batch, cfg.channels, cfg.visual_image_size,
                                                cfg.visual_image_size)), dtype=torch.float64)

    text, image = toks(), img()
    aug_t = [toks() for _ in range(n_aug_text)]
    aug_i = [img() for _ in range(n_aug_image)]
    return text, image, aug_t, aug_i


CFG1 = ClipConfig(dim_text=64, dim_image=64, dim_latent=64, num_text_tokens=1000, text_enc_depth=2,
                  text_seq_len=32, text_heads=4, visual_enc_depth=2, visual_image_size=64,
                  visual_patch_size=32, visual_heads=4)
"""BASELINE.json configs[0] (SURVEY.md 8(d) cfg1): the reference's CPU-runnable plumbing case."""
