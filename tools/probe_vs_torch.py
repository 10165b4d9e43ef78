"""The non-GEMM hot kernels beside PyTorch-ROCm's own kernels on the same data (an outside yardstick, like hipBLASLt in probe_perf.py):
attention forward + backward (b = 1024, 8 heads of 64, n = 257 / 256) against torch.nn.functional.scaled_dot_product_attention, LayerNorm
forward + backward ([263168, 512]) against F.layer_norm, the GEGLU + LayerNorm pair against its eager composition.
    python tools/probe_vs_torch.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x_clip_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16


def timeit(fn, iters=10, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def attention(n, b=1024, h=8):
    qkv = torch.randn(b, n, 3 * h * 64, device=dev, dtype=bf)
    t_f = timeit(lambda: ops.attention_fwd(qkv, None, h, 0.125))
    out, lse = ops.attention_fwd(qkv, None, h, 0.125)
    do = torch.randn_like(out)
    t_b = timeit(lambda: ops.attention_bwd(qkv, None, out, do, lse, h, 0.125))
    q, k, v = (t.contiguous().requires_grad_(True) for t in qkv.view(b, n, 3, h, 64).permute(2, 0, 3, 1, 4))   # [b, h, n, 64] each, as SDPA wants them
    r_f = timeit(lambda: F.scaled_dot_product_attention(q, k, v, scale=0.125))
    o = F.scaled_dot_product_attention(q, k, v, scale=0.125)
    go = torch.randn_like(o)

    def bwd():
        q.grad = k.grad = v.grad = None
        o.backward(go, retain_graph=True)
    r_b = timeit(bwd)
    print(f"attention n={n:3d} b={b} h={h}:  forward {t_f:7.3f} ms (torch SDPA {r_f:7.3f})   backward {t_b:7.3f} ms (torch SDPA {r_b:7.3f})   "
          f"[ours reads the packed qkv rows and writes the packed gradient; SDPA is given contiguous per-head tensors]", flush=True)


def layernorm(M=263168, D=512):
    x = torch.randn(M, D, device=dev, dtype=bf)
    g = torch.ones(D, device=dev, dtype=bf)
    t_f = timeit(lambda: ops.layernorm_fwd(x, g))
    y, mean, rstd = ops.layernorm_fwd(x, g)
    dy = torch.randn_like(y)
    t_b = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd))
    xr = x.clone().requires_grad_(True)
    gr = g.clone().requires_grad_(True)
    r_f = timeit(lambda: F.layer_norm(xr, (D,), gr, None, 1e-3))
    yr = F.layer_norm(xr, (D,), gr, None, 1e-3)

    def bwd():
        xr.grad = gr.grad = None
        yr.backward(dy, retain_graph=True)
    r_b = timeit(bwd)
    print(f"layernorm [{M}, {D}]:  forward {t_f:7.3f} ms (torch {r_f:7.3f})   backward {t_b:7.3f} ms (torch {r_b:7.3f})", flush=True)


def geglu_layernorm(M=263168, Fh=2048):
    uv = torch.randn(M, 2 * Fh, device=dev, dtype=bf)
    g = torch.ones(Fh, device=dev, dtype=bf)
    t_f = timeit(lambda: ops.layernorm_fwd(uv, g, None, True))

    def eager():
        u, t = uv.chunk(2, dim=-1)
        return F.layer_norm(u * F.gelu(t), (Fh,), g, None, 1e-3)
    r_f = timeit(eager)
    print(f"GEGLU + layernorm [{M}, {2 * Fh}] -> [{M}, {Fh}]:  forward {t_f:7.3f} ms (torch eager chunk / gelu / mul / layer_norm {r_f:7.3f})", flush=True)


if __name__ == "__main__":
    attention(257)
    attention(256)
    layernorm()
    geglu_layernorm()
