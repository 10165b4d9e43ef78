"""Host-side mirror of the reference's public interface for the contrastive-training path: `CLIP`, `TextTransformer`,
`VisionTransformer` (reference x_clip/x_clip.py:295-390,412-875; exports x_clip/__init__.py:1).

Same constructor keywords and defaults, same `forward` signature / return values / assertion messages, same `state_dict`
keys and parameter shapes, same parameter construction order (so `torch.manual_seed(s); CLIP(...)`
draws the same initial weights as the reference) -- but the modules are parameter containers: all arithmetic runs in
the gfx950 kernels behind x_clip_amd.functional / x_clip_amd.losses.  There is no CPU / ATen fallback; tensors must live
on an MI355X.

Not on this path (constructor raises NotImplementedError): the causal text encoder together with rotary embeddings, FILIP or
MLM (each fails inside the reference's own forward); dim_head > 128 (heads of up to 64
dimensions run in 64-feature head slots, 65 ... 128 in 128-feature slots, zero-padded where narrower: Transformer.stack_params).
"""
from __future__ import annotations

import copy
from typing import Optional

import torch
import torch.distributed as distributed
from torch import nn
import torch.nn.functional as F

from . import functional as XF
from . import losses as XL
from .mlm import MLM
from .visual_ssl import SimCLR, SimSiam

Tensor = torch.Tensor


def exists(val):
    return val is not None


def cast_tuple(t):
    return t if isinstance(t, (tuple, list)) else (t,)


# ---- parameter containers with the reference's attribute names ---------------------------------------------------------
class LayerNorm(nn.Module):
    """gain-only LayerNorm (x_clip.py:112-121)"""

    def __init__(self, dim):
        super().__init__()
        self.g = nn.Parameter(torch.ones(dim))


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = LayerNorm(dim)
        self.fn = fn


class GEGLU(nn.Module):
    pass


class FeedForward(nn.Module):
    """Linear(dim, 2*mult*dim) -> GEGLU -> LayerNorm -> Dropout(0) -> Linear(mult*dim, dim), no biases (x_clip.py:185-199)"""

    def __init__(self, dim, mult=4, dropout=0.):
        super().__init__()
        assert 0. <= dropout < 1.
        inner_dim = int(dim * mult)
        self.net = nn.Sequential(
            nn.Linear(dim, inner_dim * 2, bias=False),
            GEGLU(),
            LayerNorm(inner_dim),
            nn.Dropout(dropout),
            nn.Linear(inner_dim, dim, bias=False),
        )


class Attention(nn.Module):
    """fused-QKV multi-head attention parameters (x_clip.py:201-245)"""

    def __init__(self, dim, dim_head=64, heads=8, causal=False, dropout=0.):
        super().__init__()
        assert 0. <= dropout < 1.
        self.heads = heads
        self.causal = causal
        self.scale = dim_head ** -0.5
        inner_dim = dim_head * heads
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim, bias=False), LayerNorm(dim))
        self.dropout = nn.Dropout(dropout)


class Transformer(nn.Module):
    """norm_in -> depth x (pre-norm attention + skip, pre-norm feed-forward + skip) -> norm_out (x_clip.py:247-291)"""

    def __init__(self, dim, *, depth, dim_head=64, heads=8, causal=False, attn_dropout=0., ff_dropout=0., ff_mult=4,
                 checkpoint_during_training=False):
        super().__init__()
        self.checkpoint_during_training = checkpoint_during_training
        self.dim, self.depth, self.heads, self.dim_head, self.causal = dim, depth, heads, dim_head, causal
        self.attn_dropout, self.ff_dropout = float(attn_dropout), float(ff_dropout)
        XF.StackSpec(depth=depth, heads=heads, dim_head=dim_head)      # (raises for a head width the kernels do not hold)
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                PreNorm(dim, Attention(dim=dim, dim_head=dim_head, heads=heads, causal=causal, dropout=attn_dropout)),
                PreNorm(dim, FeedForward(dim=dim, mult=ff_mult, dropout=ff_dropout)),
            ]))
        self.norm_in = LayerNorm(dim)
        self.norm_out = LayerNorm(dim)
        self._pad_cache = {}                                  # narrow heads: padded weights of no-grad passes (stack_params)
        self._rotary_ok = None                                # (data_ptr, version, shape) of the last validated rotary table

    def _padded(self, w_qkv: Tensor, w_out: Tensor, dh: int, h: int, hs: int):
        """zero rows / columns up to the head slot, as differentiable views of the parameters; rebuilt only while autograd needs the
        graph (training) -- inference calls reuse the padded copies until a parameter changes (its version counter moves)"""
        def pad():
            return (F.pad(w_qkv.view(3, h, dh, -1), (0, 0, 0, hs - dh)).reshape(3 * h * hs, -1),
                    F.pad(w_out.view(-1, h, dh), (0, hs - dh)).reshape(-1, h * hs))
        if torch.is_grad_enabled() and (w_qkv.requires_grad or w_out.requires_grad):
            return pad()
        key = (w_qkv.data_ptr(), w_qkv._version, w_out.data_ptr(), w_out._version, w_qkv.dtype, w_qkv.device)
        hit = self._pad_cache.get(id(w_qkv))
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, pad())
            self._pad_cache[id(w_qkv)] = hit
        return hit[1]

    def stack_params(self):
        """flat parameter list in the order x_clip_amd.functional.stack_forward expects"""
        ps = [self.norm_in.g]
        dh, h = self.dim_head, self.heads
        for attn, ff in self.layers:
            w_qkv, w_out = attn.fn.to_qkv.weight, attn.fn.to_out[0].weight
            hs = 64 if dh <= 64 else 128                      # the kernels' head slot (StackSpec.head_slot)
            if dh < hs:                                       # heads narrower than their slot: zero rows / columns (StackSpec)
                w_qkv, w_out = self._padded(w_qkv, w_out, dh, h, hs)
            ps += [attn.norm.g, w_qkv, w_out, attn.fn.to_out[1].g,
                   ff.norm.g, ff.fn.net[0].weight, ff.fn.net[2].g, ff.fn.net[4].weight]
        ps.append(self.norm_out.g)
        return ps

    def spec(self, rotary: Optional[Tensor] = None) -> XF.StackSpec:
        # (nn.Dropout semantics: active in training mode only)
        return XF.StackSpec(depth=self.depth, heads=self.heads, dim_head=self.dim_head,
                            checkpoint=bool(self.training and self.checkpoint_during_training), rotary=rotary, causal=self.causal,
                            attn_dropout=self.attn_dropout if self.training else 0.0, ff_dropout=self.ff_dropout if self.training else 0.0)

    def forward(self, x, rotary_pos_emb=None, mask=None):
        rotary = None
        if exists(rotary_pos_emb):
            # the reference takes the [n, rot] angle table RotaryEmbedding.forward returns (rot = min(dim_head, 32), x_clip.py:166,274,311); the
            # kernels regenerate the angles from rot / 2 frequencies, which are the table's row for position 1 -- provided the table IS
            # position x frequency
            t = rotary_pos_emb.float()
            rot = t.shape[1] if t.dim() == 2 else 0
            if t.dim() != 2 or rot % 2 != 0 or not 2 <= rot <= min(32, self.dim_head) or t.shape[0] < max(2, x.shape[1]):
                raise NotImplementedError("Transformer.forward(rotary_pos_emb=table): a [n, min(dim_head, 32)] angle table covering the sequence")
            rotary = t[1, :rot // 2].contiguous()
            # the check reads the table on the host (a sync): once per table -- the same tensor, unmodified, is not checked again
            key = (rotary_pos_emb.data_ptr(), rotary_pos_emb._version, tuple(rotary_pos_emb.shape), rotary_pos_emb.dtype)
            if self._rotary_ok != key:
                want = torch.outer(torch.arange(t.shape[0], device=t.device, dtype=torch.float32), rotary).repeat(1, 2)
                if not torch.allclose(t, want, rtol=1e-5, atol=1e-6):
                    raise NotImplementedError("Transformer.forward(rotary_pos_emb=table): only tables of the form position x frequency "
                                              "(RotaryEmbedding.forward) are supported by the rotary kernel")
                self._rotary_ok = key
        return XF.transformer(x, self.stack_params(), self.spec(rotary), mask)


class RotaryEmbedding(nn.Module):
    """reference RotaryEmbedding (x_clip.py:155-166): holds the `inv_freq` buffer (state_dict key parity).  The kernels
    (xclip_rotary) regenerate the same 10000^(-2j/dim) frequencies; dim = min(dim_head, 32) (x_clip.py:311), even."""

    def __init__(self, dim):
        super().__init__()
        if dim % 2 != 0 or not 2 <= dim <= 32:
            raise NotImplementedError("rotary embedding: an even number of 2 .. 32 rotated features per head (min(dim_head, 32))")
        self.register_buffer('inv_freq', 1. / (10000 ** (torch.arange(0, dim, 2).float() / dim)))

    def frequencies(self, device) -> Tensor:
        """the dim / 2 fp32 frequencies the kernels take.  A model cast with .to(bfloat16) also rounds the buffer (the reference then
        computes its angle table from the rounded values); the frequencies are regenerated in fp32 in that case."""
        f = self.inv_freq
        if f.dtype != torch.float32:
            f = 1. / (10000 ** (torch.arange(0, 2 * f.numel(), 2, device=device).float() / (2 * f.numel())))
        return f.to(device).contiguous()

    def forward(self, seq_len, device):                          # [seq_len, dim] angle table (not used by the kernels)
        pos = torch.arange(seq_len, device=device, dtype=self.inv_freq.dtype)
        return torch.outer(pos, self.inv_freq.to(device)).repeat(1, 2)


class TextTransformer(nn.Module):
    """reference TextTransformer (x_clip.py:295-338): forward(x int64 [b, n], mask bool [b, n]) -> [b, n+1, dim]; with causal = True
    there is no CLS token (x_clip.py:314) and the output is [b, n, dim]"""

    def __init__(self, dim, *, num_tokens, max_seq_len, dim_head, rotary_pos_emb=None, causal=False, **kwargs):
        super().__init__()
        if causal and rotary_pos_emb:
            raise NotImplementedError("causal + rotary text encoder: the reference builds its angle table for n + 1 positions (x_clip.py:330) "
                                      "but a causal encoder has n (no CLS token), so its own forward fails with a shape error")
        if rotary_pos_emb and dim_head % 2 != 0:
            raise NotImplementedError("rotary embedding needs an even dim_head (the reference's rotate_half splits the rotated features in two)")
        self.token_emb = nn.Embedding(num_tokens, dim)
        self.abs_pos_emb = nn.Embedding(max_seq_len, dim) if not rotary_pos_emb else None      # x_clip.py:311-312
        self.rotary_pos_emb = RotaryEmbedding(min(dim_head, 32)) if rotary_pos_emb else None
        self.cls_token = nn.Parameter(torch.randn(dim)) if not causal else None            # x_clip.py:314
        self.transformer = Transformer(dim, dim_head=dim_head, causal=causal, **kwargs)

    def forward(self, x, mask=None, *, pool_row: Optional[int] = None):
        """pool_row = r (not a reference keyword): the caller will read row r of every sample's encoding and nothing else -- the result is
        then [b, dim], and the last layer's row-wise part (to_out, feed-forward, norm_out) runs on those rows alone (functional.stack_forward)"""
        t = self.transformer
        pos = self.abs_pos_emb.weight if exists(self.abs_pos_emb) else None
        return XF.text_encode(x, mask, self.token_emb.weight, pos, self.cls_token, t.stack_params(),
                              t.spec(rotary=self.rotary_pos_emb.frequencies(x.device) if exists(self.rotary_pos_emb) else None), pool_row=pool_row)


class PatchDropout(nn.Module):
    """keeps max(1, int(n * (1 - prob))) random patches per sample while training (x_clip.py:134-151).  Here it only
    draws the kept-index set; the gather is folded into the patchify kernel."""

    def __init__(self, prob):
        super().__init__()
        assert 0 <= prob < 1.
        self.prob = prob

    def draw(self, batch: int, n: int, device, force_keep_all=False) -> Optional[Tensor]:
        if not self.training or self.prob == 0. or force_keep_all:
            return None
        num_patches_keep = max(1, int(n * (1 - self.prob)))
        return torch.randn(batch, n, device=device).topk(num_patches_keep, dim=-1).indices.to(torch.int32)


class VisionTransformer(nn.Module):
    """reference VisionTransformer (x_clip.py:340-390): forward(image [b, c, H, W]) -> [b, 1 + n_kept, dim]"""

    def __init__(self, dim, *, image_size, patch_size, channels, patch_dropout=0.5, **kwargs):
        super().__init__()
        assert image_size % patch_size == 0, 'Image dimensions must be divisible by the patch size.'
        num_patches = (image_size // patch_size) ** 2
        patch_dim = channels * patch_size ** 2
        self.dim = dim                                           # output width (the visual-SSL projector is sized from it)
        self.patch_size = patch_size
        self.num_patches = num_patches
        # index 0 of the reference Sequential is the einops Rearrange (no parameters): keep `to_tokens.1.*` key names
        self.to_tokens = nn.Sequential(nn.Identity(), nn.Linear(patch_dim, dim))
        self.pos_emb = nn.Embedding(num_patches, dim)
        self.patch_dropout = PatchDropout(patch_dropout)
        self.transformer = Transformer(dim, **kwargs)
        self.to_cls_tokens = nn.Sequential(nn.Identity(), nn.Linear(dim, dim, bias=False), nn.Identity())
        self.keep_indices_override: Optional[Tensor] = None      # parity tests inject the PatchDropout draw here

    def forward(self, x, keep_all_patches=False, keep_indices: Optional[Tensor] = None):
        if keep_indices is None:
            keep_indices = self.keep_indices_override
        if keep_indices is None:
            keep_indices = self.patch_dropout.draw(x.shape[0], self.num_patches, x.device, keep_all_patches)
        t = self.transformer
        return XF.vision_encode(x, keep_indices, self.patch_size, self.to_tokens[1].weight, self.to_tokens[1].bias,
                                self.pos_emb.weight, self.to_cls_tokens[1].weight, t.stack_params(), t.spec())


def model_forward_with_context(*, fn, args, freeze, kwargs=None):
    """x_clip.py:394-408: a frozen encoder runs without a graph and its output is detached"""
    kwargs = kwargs or {}
    if not freeze:
        return fn(*args, **kwargs)
    with torch.no_grad():
        enc = fn(*args, **kwargs)
    return enc.detach()


class CLIP(nn.Module):
    def __init__(
        self,
        *,
        image_encoder=None,
        text_encoder=None,
        dim_text=512,
        dim_image=512,
        dim_latent=512,
        num_text_tokens=10000,
        text_enc_depth=6,
        text_seq_len=256,
        text_heads=8,
        text_dim_head=64,
        text_has_cls_token=True,
        text_pad_id=0,
        text_rotary_pos_emb=False,
        text_causal_mask=False,
        text_eos_id=None,
        text_encode_without_mask=False,
        visual_enc_depth=6,
        visual_heads=8,
        visual_dim_head=64,
        visual_image_size=256,
        visual_patch_size=32,
        visual_patch_dropout=0.5,
        visual_has_cls_token=True,
        channels=3,
        use_all_token_embeds=False,
        downsample_image_embeds=False,
        decoupled_contrastive_learning=False,
        extra_latent_projection=False,
        use_mlm=False,
        text_ssl_loss_weight=0.05,
        use_visual_ssl=False,
        visual_ssl=None,
        visual_ssl_type='simsiam',
        visual_ssl_hidden_layer=-1,
        simclr_temperature=0.1,
        image_ssl_loss_weight=0.05,
        multiview_loss_weight=0.1,
        checkpoint_during_training=False,
        sim_reg_loss_weight=0.,
        **kwargs
    ):
        super().__init__()
        assert use_all_token_embeds or (visual_has_cls_token or text_has_cls_token), 'CLS token must be included on both vision and text transformers if you are not using fine-grained contrastive learning loss'

        self.dim_text = dim_text
        self.dim_image = dim_image
        self.dim_latent = dim_latent

        self.image_channels = channels
        self.image_size = visual_image_size

        self.text_pad_id = text_pad_id
        self.text_has_cls_token = text_has_cls_token
        self.text_seq_len = text_seq_len

        self.text_encode_without_mask = text_encode_without_mask

        self.text_causal_mask = text_causal_mask
        self.text_eos_id = text_eos_id

        assert not (text_causal_mask and not exists(text_eos_id)), 'text EOS token id must be given if using causal mask in text transformer'
        if text_causal_mask and (use_all_token_embeds or use_mlm):
            # both fail inside the reference's own forward: the causal encoder has no CLS token, so `[:, 1:]` leaves n - 1 tokens against
            # an n-wide text mask (x_clip.py:705 -> size error in the FILIP einsum) / n-wide MLM labels (mlm.py:100-107)
            raise NotImplementedError("text_causal_mask together with use_all_token_embeds or use_mlm fails in the reference's own forward "
                                      "(token count n - 1 against an n-wide mask / label tensor)")

        if exists(text_encoder):
            self.text_transformer = text_encoder
        else:
            self.text_transformer = TextTransformer(
                dim=dim_text,
                num_tokens=num_text_tokens + (1 if use_mlm else 0),
                max_seq_len=text_seq_len,
                depth=text_enc_depth,
                heads=text_heads,
                causal=text_causal_mask,
                dim_head=text_dim_head,
                rotary_pos_emb=text_rotary_pos_emb,
                checkpoint_during_training=checkpoint_during_training
            )

        self.visual_has_cls_token = visual_has_cls_token

        if exists(image_encoder):
            self.visual_transformer = image_encoder
        else:
            self.visual_transformer = VisionTransformer(
                dim=dim_image,
                image_size=visual_image_size,
                patch_size=visual_patch_size,
                channels=channels,
                depth=visual_enc_depth,
                heads=visual_heads,
                dim_head=visual_dim_head,
                patch_dropout=visual_patch_dropout,
                checkpoint_during_training=checkpoint_during_training
            )

        self.use_mlm = use_mlm                                                             # x_clip.py:516-527
        self.text_ssl_loss_weight = text_ssl_loss_weight if use_mlm else 0
        if use_mlm:
            mlm_kwargs = {k[len('mlm_'):]: v for k, v in kwargs.items() if k.startswith('mlm_')}
            self.mlm = MLM(self.text_transformer, dim=dim_text, num_tokens=num_text_tokens, **mlm_kwargs)
        self.use_visual_ssl = use_visual_ssl or exists(visual_ssl)                          # x_clip.py:531-552
        self.image_ssl_loss_weight = image_ssl_loss_weight if use_visual_ssl else 0
        if self.use_visual_ssl:
            if exists(visual_ssl):
                self.visual_ssl = visual_ssl
            elif visual_ssl_type == 'simsiam':
                self.visual_ssl = SimSiam(self.visual_transformer, image_size=visual_image_size, channels=channels,
                                          hidden_layer=visual_ssl_hidden_layer)
            elif visual_ssl_type == 'simclr':
                self.visual_ssl = SimCLR(self.visual_transformer, image_size=visual_image_size, channels=channels,
                                         hidden_layer=visual_ssl_hidden_layer, temperature=simclr_temperature)
            else:
                raise ValueError(f'unknown visual_ssl_type')

        self.to_text_latent = nn.Linear(dim_text, dim_latent, bias=False)

        self.downsample_image_embeds = downsample_image_embeds
        if downsample_image_embeds:                                                        # x_clip.py:560-568
            assert use_all_token_embeds, 'must be using all token embeds for contrastive learning in order to downsampling'
            # parameter containers with the reference's Sequential indices (.1 = depthwise 4x4 / stride 2, .2 = 1x1 + bias); the
            # arithmetic runs in xclip_dwconv4s2 + xclip_gemm (functional.downsample_latents)
            self.to_visual_latent = nn.Sequential(
                nn.Identity(),
                nn.Conv2d(dim_image, dim_image, 4, stride=2, padding=1, bias=False, groups=dim_image),
                nn.Conv2d(dim_image, dim_latent, 1),
                nn.Identity())
        else:
            self.to_visual_latent = nn.Linear(dim_image, dim_latent, bias=False)

        self.temperature = nn.Parameter(torch.tensor(1.))

        self.use_all_token_embeds = use_all_token_embeds
        self.decoupled_contrastive_learning = decoupled_contrastive_learning
        self.extra_latent_projection = extra_latent_projection

        self.to_text_latent_extra = copy.deepcopy(self.to_text_latent)
        self.to_visual_latent_extra = copy.deepcopy(self.to_visual_latent)

        self.multiview_loss_weight = multiview_loss_weight

        # latched at construction like the reference (x_clip.py:591): init the process group BEFORE building the model
        self.requires_all_gather = distributed.is_available() and distributed.is_initialized() and distributed.get_world_size() > 1
        self.assume_equal_batch = False           # set True to skip the per-step batch-size exchange between ranks

        self.overlap_towers = True                 # issue the vision tower on a side stream next to the text tower (GPU only)
        # the CLS head reads one row of the text encoding (x_clip.py:708): ask the tower for that row only (forward(): `pool_row`).  Off = the
        # dense last layer, as the reference computes it (same loss and gradients either way)
        self.prune_unused_rows = True
        # >1: the text batch runs through the text tower in that many slices, each on its own HIP stream.  The tower alternates
        # MFMA-bound GEMMs with HBM-bound row kernels; with two slices in flight one slice's LayerNorm / GEGLU kernels (no LDS, few
        # registers: they fit on a CU beside a persistent GEMM work-group) run under the other slice's GEMMs.  Encoders are row
        # independent, so the result is the same; weight gradients are summed over the slices by autograd.
        self.text_micro_batches = 1
        self._micro_batch_min_rows = 64            # smaller slices than this are not worth a stream
        # >1: the image batch (all views) runs through the vision tower in that many slices ONE AFTER THE OTHER on the same stream --
        # a memory knob, not an overlap one: with checkpoint_during_training the tower keeps one input per layer for the whole batch
        # but re-materialises a full layer (qkv, the 8D-wide feed-forward activations and their gradients) inside the backward; with k
        # slices that transient is 1/k as large (ViT-L/14-336, 2 x 2048 images: 279 -> ~200 GB reserved with 2 slices).  Row
        # independent, so the result is the same; weight gradients are summed over the slices by autograd.
        self.image_micro_batches = 1
        self._streams = {}

        self.sim_reg_loss_weight = sim_reg_loss_weight
        self.has_sim_reg_loss = sim_reg_loss_weight > 0.
        if self.has_sim_reg_loss and not extra_latent_projection:
            raise ValueError("sim_reg_loss_weight > 0 needs extra_latent_projection=True: the reference's sim-reg path fails with an "
                             "einsum rank error otherwise (its *_extra latents are only reshaped under the extra projections, "
                             "x_clip.py:757-758,778)")
        if self.has_sim_reg_loss and use_all_token_embeds:
            raise NotImplementedError("sim_reg_loss_weight > 0 with use_all_token_embeds: only the CLS-latent form is on the accelerated path")

    def _side_stream(self, device, which=0):
        if device.type != "cuda":
            return None
        st = self._streams.get((device, which))
        if st is None:
            st = torch.cuda.Stream(device=device)
            self._streams[(device, which)] = st
        return st

    def _encode_text(self, text_args, freeze, kwargs=None):
        """-> list of encodings, one per slice of the batch (a single entry unless text_micro_batches > 1 applies)"""
        k, dev, b = int(self.text_micro_batches), text_args[0].device, text_args[0].shape[0]
        if k <= 1 or b % k != 0 or b // k < self._micro_batch_min_rows:
            return [model_forward_with_context(fn=self.text_transformer, args=text_args, freeze=freeze, kwargs=kwargs)]
        bs = b // k
        on_gpu = dev.type == "cuda"                              # (CPU = the test build: the slices simply run one after the other)
        main = torch.cuda.current_stream(dev) if on_gpu else None
        # narrow heads in a no-grad pass reuse padded copies of the projection weights (Transformer._pad_cache): they are built HERE, on
        # the main stream and in front of the fork event, so that a side-stream slice that hits the cache never reads them before the
        # padding kernels have run (the cache is filled by whichever slice calls stack_params first -- slice 0, after the fork)
        tt = getattr(self.text_transformer, "transformer", None)
        if on_gpu and tt is not None and hasattr(tt, "stack_params") and not (torch.is_grad_enabled() and not freeze):
            with torch.no_grad():
                tt.stack_params()
        # the fork point: everything the inputs depend on has been issued to `main` by now.  Recorded BEFORE slice 0 is issued, so the
        # side streams wait for the inputs only -- not for slice 0's kernels (a wait_stream(main) after slice 0 had been issued put every
        # later slice behind the whole of slice 0 and serialised the forward)
        fork = main.record_event() if on_gpu else None
        outs = []
        for i in range(k):
            args_i = tuple(a[i * bs: (i + 1) * bs] for a in text_args)
            if i == 0 or not on_gpu:
                outs.append(model_forward_with_context(fn=self.text_transformer, args=args_i, freeze=freeze, kwargs=kwargs))
                continue
            st = self._side_stream(dev, which=i)
            st.wait_event(fork)
            with torch.cuda.stream(st):
                outs.append(model_forward_with_context(fn=self.text_transformer, args=args_i, freeze=freeze, kwargs=kwargs))
        for i in range(1, k if on_gpu else 1):
            main.wait_stream(self._side_stream(dev, which=i))
            outs[i].record_stream(main)
        return outs

    def _encode_image(self, image, freeze):
        k, b = int(self.image_micro_batches), image.shape[0]
        if k <= 1 or b % k != 0 or not isinstance(self.visual_transformer, VisionTransformer):
            return model_forward_with_context(fn=self.visual_transformer, args=(image,), freeze=freeze)
        bs = b // k
        vt = self.visual_transformer
        keep_all = vt.keep_indices_override
        outs = []
        try:
            for i in range(k):
                if keep_all is not None:                         # an injected PatchDropout draw is per image: slice it too
                    vt.keep_indices_override = keep_all[i * bs: (i + 1) * bs]
                outs.append(model_forward_with_context(fn=vt, args=(image[i * bs: (i + 1) * bs],), freeze=freeze))
        finally:
            vt.keep_indices_override = keep_all
        return torch.cat(outs, dim=0)

    def forward(
        self,
        text,
        image,
        return_loss=False,
        return_encodings=False,
        return_latents=False,
        freeze_image_encoder=False,
        freeze_text_encoder=False,
        text_to_image=True,
        aug_text=None,
        aug_image=None
    ):
        batch, device = text.shape[0], text.device

        text_mask = text != self.text_pad_id                                               # x_clip.py:614

        text_ssl_loss = 0                                                                  # x_clip.py:618-622
        if return_loss and self.use_mlm:
            text_ssl_loss = self.mlm(text, mask=text_mask)
        image_ssl_loss = 0
        if return_loss and self.use_visual_ssl:
            image_ssl_loss = self.visual_ssl(image)

        num_batch_texts = num_batch_images = 1

        if exists(aug_text):                                                               # x_clip.py:629-639
            aug_text = cast_tuple(aug_text)
            assert all(map(lambda t: t.shape == text.shape, aug_text))
            num_batch_texts = len(aug_text) + 1
            aug_text = torch.cat(aug_text, dim=0)
            aug_text_mask = aug_text != self.text_pad_id
            text_mask = torch.cat((text_mask, aug_text_mask), dim=0)
            text = torch.cat((text, aug_text), dim=0)

        if exists(aug_image):                                                              # x_clip.py:641-648
            aug_image = cast_tuple(aug_image)
            assert all(map(lambda i: i.shape == image.shape, aug_image))
            num_batch_images = len(aug_image) + 1
            aug_image = torch.cat(aug_image, dim=0)
            image = torch.cat((image, aug_image), dim=0)

        is_multiview = (num_batch_texts > 1 or num_batch_images > 1)
        assert not (return_loss and not self.training), 'loss cannot be used if not training'
        assert not (not return_loss and is_multiview), 'do not pass in augmented texts or images if not training'
        assert not (self.multiview_loss_weight == 0 and is_multiview), 'multiview loss weight cannot be 0 if augmented text or images passed in'

        text_args = (text,)
        if not self.text_encode_without_mask:
            text_args = (*text_args, text_mask)
        # x_clip.py:708: the CLS path reads `enc_text[:, 0]` and nothing else -- the text tower is then asked for that row only and runs the
        # row-wise part of its last layer on it (functional.stack_forward `pool_row`; the same loss and gradients up to bf16 rounding order: the other rows of that part
        # are dead in the forward and their gradient is exactly zero in the backward).  `prune_unused_rows = False` keeps the dense last layer.
        text_kwargs = None
        if (self.prune_unused_rows and isinstance(self.text_transformer, TextTransformer) and self.text_has_cls_token
                and not self.text_causal_mask and not self.use_all_token_embeds and not return_encodings):
            text_kwargs = dict(pool_row=0)

        # The two towers are independent until the head.  On a GPU the vision tower is issued on a side HIP stream (autograd
        # replays its backward there too), so its small kernels fill the gaps the text tower's leaves; the head waits for both.
        side = self._side_stream(image.device) if self.overlap_towers else None
        if side is not None:
            main = torch.cuda.current_stream(image.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                enc_image = self._encode_image(image, freeze_image_encoder)
            enc_text_parts = self._encode_text(text_args, freeze_text_encoder, text_kwargs)
            main.wait_stream(side)
            enc_image.record_stream(main)
        else:
            enc_text_parts = self._encode_text(text_args, freeze_text_encoder, text_kwargs)
            enc_image = self._encode_image(image, freeze_image_encoder)
        # (the CLS path below only needs row 0 of every sample: the slices are joined after that selection, not before)
        cls_only = (len(enc_text_parts) > 1 and not self.text_causal_mask and not return_encodings and not self.use_all_token_embeds
                    and enc_text_parts[0].ndim == 3)
        enc_text = enc_text_parts[0] if (len(enc_text_parts) == 1 or cls_only) else torch.cat(enc_text_parts, dim=0)

        if self.text_causal_mask:                                                          # x_clip.py:670-685 (its `b` is the batch size)
            enc_text = XF.eos_to_front(enc_text, text, self.text_eos_id)

        if return_encodings:                                                               # x_clip.py:697-698
            return enc_text, enc_image

        if self.use_all_token_embeds:                                                      # x_clip.py:702-706
            assert enc_text.ndim == 3, 'encoded text must have 3 dimensions (batch, seq, features)'
            assert enc_image.ndim == 3, 'encoded image must have 3 dimensions (batch, seq [height x width], features)'
            text_embeds = enc_text[:, 1:] if self.text_has_cls_token else enc_text
            image_embeds = enc_image[:, 1:] if self.visual_has_cls_token else enc_image
        else:                                                                              # x_clip.py:708-709
            if cls_only:
                text_embeds = torch.cat([XF.select_row(e, 0) for e in enc_text_parts], dim=0)
            else:
                text_embeds = XF.select_row(enc_text, 0) if enc_text.ndim == 3 else enc_text
            image_embeds = XF.select_row(enc_image, 0) if enc_image.ndim == 3 else enc_image

        text_latents = XF.l2norm(XF.linear(text_embeds, self.to_text_latent.weight))       # x_clip.py:713-715
        def visual_latents(proj):
            if self.downsample_image_embeds:
                return XF.downsample_latents(image_embeds, proj[1].weight, proj[2].weight, proj[2].bias)
            return XF.linear(image_embeds, proj.weight)

        image_latents = XF.l2norm(visual_latents(self.to_visual_latent))

        text_latents_extra, image_latents_extra = text_latents, image_latents              # x_clip.py:720-724
        if self.extra_latent_projection:
            text_latents_extra = XF.l2norm(XF.linear(text_embeds, self.to_text_latent_extra.weight))
            image_latents_extra = XF.l2norm(visual_latents(self.to_visual_latent_extra))

        if return_latents:                                                                 # x_clip.py:728-732
            if self.extra_latent_projection:
                return text_latents, image_latents, text_latents_extra, image_latents_extra
            return text_latents, image_latents

        if not return_loss:                                                                # x_clip.py:740-746 (inference only)
            temp = self.temperature.exp()
            a, b = (text_latents_extra, image_latents_extra) if self.extra_latent_projection and not text_to_image \
                else (text_latents, image_latents)
            if self.use_all_token_embeds:
                return XF.token_similarity(a, b) * temp                                    # einsum('b t d, b i d -> b t i') * temp
            return XF.pair_similarity(a, b) * temp                                         # einsum('b d, b d -> b') * temp

        # ---- training loss -----------------------------------------------------------------------------------------------
        def split_views(t, m):                                                             # '(m b) ... -> m b ...'
            return t.reshape(m, t.shape[0] // m, *t.shape[1:])

        text_latents = split_views(text_latents, num_batch_texts)
        image_latents = split_views(image_latents, num_batch_images)
        if self.extra_latent_projection:
            text_latents_extra = split_views(text_latents_extra, num_batch_texts)
            image_latents_extra = split_views(image_latents_extra, num_batch_images)

        multiview_loss_weight = self.multiview_loss_weight if is_multiview else 0          # x_clip.py:851-855
        cl_loss_weight = 1 - (self.text_ssl_loss_weight + self.image_ssl_loss_weight + multiview_loss_weight)

        spec = XL.ContrastiveSpec(dcl=self.decoupled_contrastive_learning, main_weight=cl_loss_weight,
                                  multiview_weight=multiview_loss_weight, distributed=self.requires_all_gather,
                                  assume_equal_batch=self.assume_equal_batch)
        if self.use_all_token_embeds:                                                      # x_clip.py:797-811
            loss = XL.filip_loss(self.temperature, text_latents, image_latents,
                                 text_latents_extra if self.extra_latent_projection else None,
                                 image_latents_extra if self.extra_latent_projection else None, text_mask, spec)
        else:
            loss = XL.contrastive_loss(self.temperature, text_latents, image_latents,
                                       text_latents_extra if self.extra_latent_projection else None,
                                       image_latents_extra if self.extra_latent_projection else None, spec)
        if self.use_mlm:                                                                   # x_clip.py:857-860
            loss = loss + text_ssl_loss * self.text_ssl_loss_weight
        if self.use_visual_ssl:
            loss = loss + image_ssl_loss * self.image_ssl_loss_weight
        if self.has_sim_reg_loss:                                                          # x_clip.py:773-784, 872-873
            assert not is_multiview, 'the similarity regularisation loss is defined for a single view (its [1, b, b] mask, x_clip.py:776-778)'
            sim_reg = XL.sim_reg_loss(text_latents[0], image_latents[0], text_latents_extra[0], image_latents_extra[0], spec)
            loss = loss + sim_reg * self.sim_reg_loss_weight
        return loss
