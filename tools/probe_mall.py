"""Does producer -> consumer chunking keep the FF hidden activation in the 256 MB Infinity Cache?  FF1 GEMM + GEGLU-LN forward over the
whole [263168, 4096] activation vs in row chunks (the consumer reads what the producer just wrote)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_clip_amd import ops
dev = torch.device("cuda")
M, D, F2 = 1024 * 257, 512, 4096
h2 = torch.randn(M, D, device=dev, dtype=torch.bfloat16)
w = torch.randn(F2, D, device=dev, dtype=torch.bfloat16) * 0.05
g = torch.ones(F2 // 2, device=dev, dtype=torch.bfloat16)
u = torch.empty(M, F2, device=dev, dtype=torch.bfloat16)


def run(chunk):
    for c0 in range(0, M, chunk):
        c1 = min(M, c0 + chunk)
        ops.gemm(h2[c0:c1], w, c1 - c0, F2, D, out=u[c0:c1])
        ops.layernorm_fwd(u[c0:c1], g, None, True)


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for chunk in (M, 131584, 65792, 32896, 16448, 8224):
    print(f"chunk {chunk:7d} rows ({chunk * F2 * 2 / 1e6:7.1f} MB of u): FF1 + GEGLU-LN fwd {timeit(lambda: run(chunk)):8.1f} us", flush=True)
