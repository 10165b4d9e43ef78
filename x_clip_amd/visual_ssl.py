"""Visual self-supervision side loss: the host mirror of reference x_clip/visual_ssl.py (`SimSiam`, visual_ssl.py:207-259, with
its `NetWrapper` :141-203 and the BatchNorm MLPs :112-136).

`SimSiam(net, image_size, ...)` takes the image encoder (the CLIP vision tower itself: shared parameters, four more encoder
passes per step -- two views through the online branch, the same two through the stop-gradient target branch) and returns
the symmetric negative-cosine loss.  Everything that touches activations runs in the gfx950 kernels: the encoder passes are
the accelerated VisionTransformer, the projector / predictor Linear layers are xclip_gemm, BatchNorm1d + ReLU is
xclip_batchnorm_fwd / _bwd (column statistics, running-statistics update included) and the loss is xclip_neg_cosine_fwd / _bwd.
The modules are parameter containers with the reference's Sequential indices, so `state_dict` keys and shapes match.

Host-side pieces that stay in torch, as in the reference: the two augmentation callables.  The default pipeline
(visual_ssl.py:24-45) is torchvision's; it is imported lazily, and when torchvision is not installed `augment_fn` must be
given (the reference has the same dependency at import time).

Differences from the reference, all at construction time: the lazily built projector (visual_ssl.py:167-171) is created in the
constructor when the encoder's output width is known (`representation_dim=` or an encoder with a `.dim` attribute) instead of
by a mock forward on random data (visual_ssl.py:235) -- the kernels need device tensors and a model is usually built on the host
first; otherwise it is created on the first forward, exactly like the reference's singleton.

`SimCLR(net, image_size, ...)` (visual_ssl.py:263-299) is the NT-Xent variant: two views through the same wrapper, loss over the
2R x 2R projection similarities with the diagonal removed -- computed block-wise by the contrastive head's kernels.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn
from torch.autograd.function import once_differentiable

from . import functional as XF
from . import ops

Tensor = torch.Tensor


def default(val, def_val):
    return def_val if val is None else val


class RandomApply(nn.Module):
    """visual_ssl.py:13-21"""

    def __init__(self, fn, p):
        super().__init__()
        self.fn = fn
        self.p = p

    def forward(self, x):
        import random
        if random.random() > self.p:
            return x
        return self.fn(x)


def get_default_aug(image_size, channels=3):
    """the reference's default SimCLR-style augmentation (visual_ssl.py:23-45); needs torchvision on the host"""
    try:
        from torchvision import transforms as T
    except ImportError as e:                                   # pragma: no cover - depends on the host image
        raise ImportError("the default visual-SSL augmentation pipeline is torchvision's (reference visual_ssl.py:24-45) and "
                          "torchvision is not installed: pass augment_fn= (and augment_fn2=) to SimSiam, or a ready "
                          "visual_ssl= module to CLIP") from e
    is_rgb = channels == 3
    is_greyscale = channels == 1
    rgb_or_greyscale = is_rgb or is_greyscale
    return torch.nn.Sequential(
        RandomApply(T.ColorJitter(0.8, 0.8, 0.8, 0.2), p=0.3) if rgb_or_greyscale else nn.Identity(),
        T.RandomGrayscale(p=0.2) if is_rgb else nn.Identity(),
        T.RandomHorizontalFlip(),
        RandomApply(T.GaussianBlur((3, 3), (1.0, 2.0)), p=0.2),
        T.RandomResizedCrop((image_size, image_size)),
        T.Normalize(mean=torch.tensor([0.485, 0.456, 0.406]), std=torch.tensor([0.229, 0.224, 0.225])) if is_rgb else nn.Identity(),
    )


# ---- differentiable pieces -------------------------------------------------------------------------------------------------
class _LinearBiasFn(torch.autograd.Function):
    """y = x W^T + b on [rows, K] (nn.Linear with bias: the predictor MLP, visual_ssl.py:112-120)"""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, b: Tensor):
        x = ops._c(x)
        M, K = x.shape
        N = w.shape[0]
        y = ops.gemm(x, ops._c(w), M, N, K, bias=ops._c(b))
        ctx.save_for_backward(x, w)
        ctx.bdt = b.dtype
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        M, K = x.shape
        N = w.shape[0]
        dy = ops._c(dy)
        dx = ops.gemm(dy, ops._c(w), M, K, N, b_kmajor=True) if ctx.needs_input_grad[0] else None
        dw = ops.gemm(dy, x, N, K, M, a_kmajor=True, b_kmajor=True) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.needs_input_grad[2]:
            acc = torch.zeros(N, dtype=torch.float32, device=dy.device)
            ops.rows_scatter_add(dy, None, None, acc)
            db = acc.to(ctx.bdt)
        return dx, dw, db


class _BatchNormFn(torch.autograd.Function):
    """BatchNorm1d (+ the ReLU that follows it) over the rows of x [rows, C].  running_mean / running_var are fp32 work copies
    updated in place by the kernel (the module copies them back into its buffers)."""

    @staticmethod
    def forward(ctx, x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], running_mean, running_var, momentum: float, eps: float,
                training: bool, relu: bool):
        x = ops._c(x)
        g32, b32 = ops._f32(gamma), ops._f32(beta)
        y, mean, rstd = ops.batchnorm_fwd(x, g32, b32, running_mean, running_var, momentum, eps, training, relu)
        ctx.save_for_backward(x, g32, b32, mean, rstd)
        ctx.meta = (training, relu, None if gamma is None else gamma.dtype, None if beta is None else beta.dtype)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, g32, b32, mean, rstd = ctx.saved_tensors
        training, relu, gdt, bdt = ctx.meta
        affine = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx, dg, db = ops.batchnorm_bwd(x, dy, g32, b32, mean, rstd, training, relu, affine)
        return (dx if ctx.needs_input_grad[0] else None, dg.to(gdt) if ctx.needs_input_grad[1] else None,
                db.to(bdt) if ctx.needs_input_grad[2] else None, None, None, None, None, None, None)


class _NegCosineFn(torch.autograd.Function):
    """coef sum_r (2 - 2 cos(p_r, z_r)) (fp32 scalar); z is the stop-gradient target (visual_ssl.py:104-107,243-256)"""

    @staticmethod
    def forward(ctx, p: Tensor, z: Tensor, coef: float):
        p, z = ops._c(p), ops._c(z.detach())
        acc = torch.zeros(1, dtype=torch.float32, device=p.device)
        st = ops.neg_cosine_fwd(p, z, coef, acc)
        ctx.save_for_backward(p, z, *st)
        ctx.coef = coef
        return acc.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        p, z, c, rp, rz = ctx.saved_tensors
        g = dloss.detach().float().reshape(1).contiguous()
        return ops.neg_cosine_bwd(p, z, (c, rp, rz), g, ctx.coef), None, None


# ---- modules ---------------------------------------------------------------------------------------------------------------
class BatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d's parameters / buffers / state_dict, the kernels' arithmetic.  `forward(x, relu=False)`, x [rows, C]."""

    def forward(self, x: Tensor, relu: bool = False) -> Tensor:
        assert x.dim() == 2 and x.shape[1] == self.num_features, 'expected a [rows, num_features] input'
        use_batch_stats = self.training or self.running_mean is None
        factor = 0.0 if self.momentum is None else self.momentum
        rm = rv = None
        if self.running_mean is not None:
            if self.training and self.num_batches_tracked is not None:
                self.num_batches_tracked.add_(1)
                if self.momentum is None:                      # cumulative moving average
                    factor = 1.0 / float(self.num_batches_tracked)
            # fp32 work copies only when the buffers were cast (model.to(bfloat16))
            rm = self.running_mean if self.running_mean.dtype == torch.float32 else self.running_mean.float()
            rv = self.running_var if self.running_var.dtype == torch.float32 else self.running_var.float()
        if use_batch_stats:
            assert x.shape[0] > 1, 'Expected more than 1 value per channel when training'
        y = _BatchNormFn.apply(x, self.weight, self.bias, rm, rv, factor, self.eps, use_batch_stats, relu)
        if self.training and rm is not None and rm is not self.running_mean:
            self.running_mean.copy_(rm)
            self.running_var.copy_(rv)
        return y


class _MLP(nn.Sequential):
    """a Sequential of Linear / BatchNorm1d / ReLU in the reference's order (so the state_dict indices match); the forward pass
    walks it and gives every BatchNorm1d the ReLU that follows it"""

    def forward(self, x: Tensor) -> Tensor:
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear):
                x = XF.linear(x, m.weight) if m.bias is None else _LinearBiasFn.apply(x, m.weight, m.bias)
            elif isinstance(m, BatchNorm1d):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = m(x, relu=relu)
                i += int(relu)
            else:
                raise TypeError(f'unexpected layer {type(m).__name__} in the SSL MLP')
            i += 1
        return x


def MLP(dim, projection_size, hidden_size=None):
    """predictor (visual_ssl.py:112-120)"""
    hidden_size = default(hidden_size, dim)
    return _MLP(nn.Linear(dim, hidden_size), BatchNorm1d(hidden_size), nn.ReLU(inplace=True), nn.Linear(hidden_size, projection_size))


def SimSiamMLP(dim, projection_size, hidden_size=4096):
    """projector (visual_ssl.py:122-135)"""
    hidden_size = default(hidden_size, projection_size * 2)
    return _MLP(nn.Linear(dim, hidden_size, bias=False), BatchNorm1d(hidden_size), nn.ReLU(inplace=True),
                nn.Linear(hidden_size, hidden_size, bias=False), BatchNorm1d(hidden_size), nn.ReLU(inplace=True),
                nn.Linear(hidden_size, projection_size, bias=False), BatchNorm1d(projection_size, affine=False))


class NetWrapper(nn.Module):
    """visual_ssl.py:141-203: runs the encoder, takes the representation at `layer` (-1: the encoder output itself; otherwise a
    forward hook on a child module -- that only fires for encoders whose forward CALLS their children, i.e. user encoders, not
    the fused VisionTransformer) and projects its flattened rows"""

    def __init__(self, net, projection_size, projection_hidden_size=4096, layer=-2, representation_dim: Optional[int] = None):
        super().__init__()
        self.net = net
        self.layer = layer
        self.projector = None
        self.projection_size = projection_size
        self.projection_hidden_size = projection_hidden_size
        self.hidden = {}
        self.hook_registered = False
        if representation_dim is not None:
            self.projector = SimSiamMLP(representation_dim, projection_size, projection_hidden_size)

    def _find_layer(self):
        if type(self.layer) == str:
            return dict([*self.net.named_modules()]).get(self.layer, None)
        if type(self.layer) == int:
            return [*self.net.children()][self.layer]
        return None

    def _hook(self, _, input, output):
        self.hidden[input[0].device] = output.reshape(output.shape[0], -1)

    def _register_hook(self):
        layer = self._find_layer()
        assert layer is not None, f'hidden layer ({self.layer}) not found'
        layer.register_forward_hook(self._hook)
        self.hook_registered = True

    def _get_projector(self, hidden: Tensor):
        if self.projector is None:
            self.projector = SimSiamMLP(hidden.shape[1], self.projection_size, self.projection_hidden_size).to(hidden)
        return self.projector

    def get_representation(self, x):
        if self.layer == -1:
            return self.net(x)
        if not self.hook_registered:
            self._register_hook()
        self.hidden.clear()
        _ = self.net(x)
        assert x.device in self.hidden, (f'hidden layer {self.layer} never emitted an output (the fused VisionTransformer does not call '
                                         f'its child modules: use hidden_layer = -1, the CLIP default)')
        hidden = self.hidden[x.device]
        self.hidden.clear()
        return hidden

    def forward(self, x, return_projection=True):
        representation = self.get_representation(x)
        if not return_projection:
            return representation
        flat = representation.reshape(-1, representation.shape[-1])            # '... d -> (...) d'
        projector = self._get_projector(flat)
        return projector(flat), representation


class SimSiam(nn.Module):
    """reference SimSiam (visual_ssl.py:207-259): forward(image [b, c, H, W]) -> scalar loss"""

    def __init__(self, net, image_size, channels=3, hidden_layer=-2, projection_size=256, projection_hidden_size=4096, augment_fn=None,
                 augment_fn2=None, representation_dim: Optional[int] = None):
        super().__init__()
        self.net = net
        self.augment1 = augment_fn if augment_fn is not None else get_default_aug(image_size, channels)
        self.augment2 = default(augment_fn2, self.augment1)
        if representation_dim is None and hidden_layer == -1:
            representation_dim = getattr(net, 'dim', None)
        self.online_encoder = NetWrapper(net, projection_size, projection_hidden_size, layer=hidden_layer, representation_dim=representation_dim)
        self.online_predictor = MLP(projection_size, projection_size, projection_hidden_size)
        params = list(net.parameters())
        if params:
            self.to(params[0].device)

    def forward(self, x):
        assert not (self.training and x.shape[0] == 1), 'you must have greater than 1 sample when training, due to the batchnorm in the projection layer'
        image_one, image_two = self.augment1(x), self.augment2(x)
        online_proj_one, _ = self.online_encoder(image_one)
        online_proj_two, _ = self.online_encoder(image_two)
        online_pred_one = self.online_predictor(online_proj_one)
        online_pred_two = self.online_predictor(online_proj_two)
        with torch.no_grad():                                                  # the target encoder IS the online encoder (visual_ssl.py:243-249)
            target_proj_one, _ = self.online_encoder(image_one)
            target_proj_two, _ = self.online_encoder(image_two)
        rows = online_pred_one.shape[0]
        # (loss_one + loss_two).mean() over the rows                              visual_ssl.py:251-259
        return (_NegCosineFn.apply(online_pred_one, target_proj_two, 1.0 / rows)
                + _NegCosineFn.apply(online_pred_two, target_proj_one, 1.0 / rows))


class _NtXentFn(torch.autograd.Function):
    """nt_xent_loss (visual_ssl.py:90-102) of queries / keys [R, d]: with P = [queries; keys] and N = 2R, the mean over the N rows of
    -log softmax over the row's N - 1 off-diagonal logits P P^T / temperature at its partner's column.  The N x N logits are never
    formed: the four R x R blocks go through the contrastive head's kernels (xclip_simloss_partial / _combine for the row
    log-sum-exps, xclip_simloss_grad for the gradient factors, whose a / c terms give G + G^T of a symmetric block in one pass)."""

    @staticmethod
    def forward(ctx, queries: Tensor, keys: Tensor, temperature: float):
        Q, K = ops._c(queries), ops._c(keys)
        R, d = Q.shape
        v = ops.vec(Q.dtype)
        if d % v:                                              # the kernels take whole 16-byte chunks: zero columns change no dot product
            Q = torch.nn.functional.pad(Q, (0, v - d % v))
            K = torch.nn.functional.pad(K, (0, v - d % v))
        acc = torch.zeros(1, dtype=torch.float32, device=Q.device)
        scale, coef = 1.0 / temperature, 1.0 / (2 * R)
        lse_q = ops.ntxent_lse(Q, K, scale, coef, acc)
        lse_k = ops.ntxent_lse(K, Q, scale, coef, acc)
        ctx.save_for_backward(Q, K, lse_q, lse_k)
        ctx.meta = (scale, coef, d, queries.dtype)
        return acc.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss):
        Q, K, lse_q, lse_k = ctx.saved_tensors
        scale, coef, d, dt = ctx.meta
        R, dp = Q.shape
        g = dloss.detach().float().reshape(1).contiguous()
        # self blocks (diagonal excluded, no positive): G + G^T in one pass; cross block: both row softmaxes and the two positives
        Gq = ops.simloss_grad(Q, Q, scale, 0, True, coef, coef, 0.0, lse_q, lse_q, None, gmul=g, times_scale=True)
        Gk = ops.simloss_grad(K, K, scale, 0, True, coef, coef, 0.0, lse_k, lse_k, None, gmul=g, times_scale=True)
        Gx = ops.simloss_grad(Q, K, scale, 0, False, coef, coef, 2 * coef, lse_q, lse_k, None, gmul=g, times_scale=True)
        dQ = ops.gemm(Gq[:, :R], Q, R, dp, R, b_kmajor=True)
        dQ = ops.gemm(Gx[:, :R], K, R, dp, R, b_kmajor=True, residual=dQ, out=dQ)
        dK = ops.gemm(Gk[:, :R], K, R, dp, R, b_kmajor=True)
        dK = ops.gemm(Gx[:, :R], Q, R, dp, R, a_kmajor=True, b_kmajor=True, residual=dK, out=dK)
        return dQ[:, :d].to(dt), dK[:, :d].to(dt), None


def nt_xent_loss(queries: Tensor, keys: Tensor, temperature: float = 0.1) -> Tensor:
    return _NtXentFn.apply(queries, keys, temperature)


class SimCLR(nn.Module):
    """reference SimCLR (visual_ssl.py:263-299): forward(image [b, c, H, W]) -> NT-Xent loss between the projections of two augmented
    views; every token row of the representation is a sample (NetWrapper flattens '... d -> (...) d').  `augment_fn` is called twice
    per step (once per view).  Note the reference divides the raw, un-normalised projections' dot products by the temperature."""

    def __init__(self, net, image_size, channels=3, hidden_layer=-2, project_hidden=True, project_dim=128, augment_both=True,
                 use_nt_xent_loss=False, augment_fn=None, temperature=0.1, representation_dim: Optional[int] = None):
        super().__init__()
        if representation_dim is None and hidden_layer == -1:
            representation_dim = getattr(net, 'dim', None)
        self.net = NetWrapper(net, project_dim, layer=hidden_layer, representation_dim=representation_dim)
        self.augment = augment_fn if augment_fn is not None else get_default_aug(image_size, channels)
        self.augment_both = augment_both
        self.temperature = temperature
        params = list(net.parameters())
        if params:
            self.to(params[0].device)

    def forward(self, x):
        transform_fn = self.augment if self.augment_both else (lambda t: t)     # (the reference names an undefined `noop` here)
        queries, _ = self.net(transform_fn(x))
        keys, _ = self.net(self.augment(x))
        queries, keys = queries.reshape(queries.shape[0], -1), keys.reshape(keys.shape[0], -1)
        return nt_xent_loss(queries, keys, temperature=self.temperature)
