"""What the reference's WAY of computing the step costs on this GPU: the default CLIP architecture (BASELINE configs[1]: dim 512, depth 6 / 6, 8 heads of
64, text 256 tokens + CLS, 256 x 256 images in 32 x 32 patches with patch dropout 0.5, InfoNCE) written as plain eager PyTorch-ROCm modules with the
operations lucidrains/x-clip uses -- nn.Linear, an einsum attention with an fp32 softmax, LayerNorm from var / mean with a gain only, chunk + gelu
GEGLU with a LayerNorm inside the feed-forward, pre-norm residual blocks (x_clip.py:111-121,180-199,201-245,274-291) -- forward + backward, bf16,
b = 1024, beside this repository's step on the same box.  It is NOT the reference (which is absent from the GPU box) and not the oracle: an independent
few-line model of the same shapes, a yardstick for "the same architecture through the framework's stock kernels".
    python tools/probe_eager_torch_step.py [batch] [--fused]      (--fused: F.layer_norm and scaled_dot_product_attention in place of the elementwise forms)"""
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")


STOCK_FUSED = False        # --fused: the framework's fused kernels where it has them (F.layer_norm, scaled_dot_product_attention) instead of the
                           # reference's elementwise formulations: the best a user gets from stock PyTorch-ROCm without writing kernels


class Norm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.g = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        if STOCK_FUSED:
            return F.layer_norm(x, x.shape[-1:], self.g, None, 1e-3)
        var = torch.var(x, dim=-1, unbiased=False, keepdim=True)
        mean = torch.mean(x, dim=-1, keepdim=True)
        return (x - mean) * (var + 1e-3).rsqrt() * self.g


class Attn(nn.Module):
    def __init__(self, dim, heads=8, dh=64):
        super().__init__()
        self.h, self.scale = heads, dh ** -0.5
        self.norm = Norm(dim)
        self.qkv = nn.Linear(dim, 3 * heads * dh, bias=False)
        self.out = nn.Linear(heads * dh, dim, bias=False)
        self.out_norm = Norm(dim)

    def forward(self, x, mask=None):
        b, n, _ = x.shape
        q, k, v = self.qkv(self.norm(x)).view(b, n, 3, self.h, -1).permute(2, 0, 3, 1, 4)
        if STOCK_FUSED:
            am = None if mask is None else mask[:, None, None, :]
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=am, scale=self.scale).permute(0, 2, 1, 3).reshape(b, n, -1)
            return self.out_norm(self.out(o))
        sim = torch.einsum("bhid,bhjd->bhij", q * self.scale, k)
        if mask is not None:
            sim = sim.masked_fill(~mask[:, None, None, :], -torch.finfo(sim.dtype).max)
        attn = sim.softmax(dim=-1, dtype=torch.float32).to(sim.dtype)
        o = torch.einsum("bhij,bhjd->bhid", attn, v).permute(0, 2, 1, 3).reshape(b, n, -1)
        return self.out_norm(self.out(o))


class FF(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.norm = Norm(dim)
        self.w1 = nn.Linear(dim, 2 * mult * dim, bias=False)
        self.mid = Norm(mult * dim)
        self.w2 = nn.Linear(mult * dim, dim, bias=False)

    def forward(self, x):
        u, t = self.w1(self.norm(x)).chunk(2, dim=-1)
        return self.w2(self.mid(u * F.gelu(t)))


class Stack(nn.Module):
    def __init__(self, dim, depth):
        super().__init__()
        self.norm_in, self.norm_out = Norm(dim), Norm(dim)
        self.layers = nn.ModuleList([nn.ModuleList([Attn(dim), FF(dim)]) for _ in range(depth)])

    def forward(self, x, mask=None):
        x = self.norm_in(x)
        for a, f in self.layers:
            x = a(x, mask) + x
            x = f(x) + x
        return self.norm_out(x)


class EagerCLIP(nn.Module):
    def __init__(self, dim=512, depth=6, vocab=10000, seq=256, image=256, patch=32, keep=0.5):
        super().__init__()
        self.patch, self.keep = patch, keep
        self.tok, self.pos = nn.Embedding(vocab, dim), nn.Embedding(seq, dim)
        self.cls = nn.Parameter(torch.randn(dim))
        self.text = Stack(dim, depth)
        npatch = (image // patch) ** 2
        self.embed = nn.Linear(3 * patch * patch, dim)
        self.ipos = nn.Embedding(npatch, dim)
        self.vision = Stack(dim, depth)
        self.to_cls = nn.Linear(dim, dim, bias=False)
        self.t_lat, self.i_lat = nn.Linear(dim, dim, bias=False), nn.Linear(dim, dim, bias=False)
        self.temp = nn.Parameter(torch.tensor(1.0))

    def forward(self, text, image):
        b = text.shape[0]
        mask = F.pad(text != 0, (1, 0), value=True)
        x = self.tok(text) + self.pos(torch.arange(text.shape[1], device=text.device))
        x = torch.cat([self.cls.expand(b, 1, -1), x], dim=1)
        te = self.text(x, mask)[:, 0]
        p = self.patch
        y = image.unfold(2, p, p).unfold(3, p, p).permute(0, 2, 3, 4, 5, 1).reshape(b, -1, p * p * 3)
        y = self.embed(y) + self.ipos(torch.arange(y.shape[1], device=y.device))
        nk = max(1, int(y.shape[1] * self.keep))
        idx = torch.randn(b, y.shape[1], device=y.device).topk(nk, dim=-1).indices
        y = y[torch.arange(b, device=y.device)[:, None], idx]
        ie = self.to_cls(self.vision(y).mean(dim=1))
        tl, il = F.normalize(self.t_lat(te), dim=-1), F.normalize(self.i_lat(ie), dim=-1)
        sim = (tl @ il.t()).float() * self.temp.exp()
        lab = torch.arange(b, device=sim.device)
        return 0.5 * (F.cross_entropy(sim, lab) + F.cross_entropy(sim.t(), lab))


def main():
    global STOCK_FUSED
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    STOCK_FUSED = "--fused" in sys.argv
    b = int(argv[0]) if argv else 1024
    torch.manual_seed(0)
    text = torch.randint(1, 10000, (b, 256), device=dev)
    image = torch.randn(b, 3, 256, 256, device=dev, dtype=torch.bfloat16)

    def timed(step, label, warm=2, iters=5):
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        print(f"{label}: {ms:8.2f} ms per step = {b / ms * 1e3:9.0f} pairs/s   (peak reserved {torch.cuda.max_memory_reserved() / 2**30:.1f} GiB)", flush=True)
        return ms

    m = EagerCLIP().to(torch.bfloat16).to(dev).train()

    def eager_step():
        m.zero_grad(set_to_none=True)
        m(text, image).backward()
    e = timed(eager_step, f"eager PyTorch-ROCm modules{' (F.layer_norm + SDPA)' if STOCK_FUSED else ' (the reference`s formulations)'}, b = {b}, bf16")
    del m
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()

    from x_clip_amd import CLIP
    c = CLIP().to(torch.bfloat16).to(dev).train()

    def our_step():
        c.zero_grad(set_to_none=True)
        c(text, image, return_loss=True).backward()
    o = timed(our_step, f"x_clip_amd.CLIP (this repository), b = {b}, bf16   ")
    print(f"ratio: {e / o:.2f} x")


if __name__ == "__main__":
    main()
