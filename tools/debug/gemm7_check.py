"""XCLIP_GEMM=7 python tools/debug/gemm7_check.py : the four-waves-of-128x128 kernel (gemm7.h) against torch on shapes it takes, then its
time on the text tower's forward shapes (run without XCLIP_GEMM for the production kernel's)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from x_clip_amd import ops
dev = torch.device("cuda:0")
bad = 0
for (M, N, K, alpha) in [(512, 256, 128, 1.0), (2048, 1536, 512, 1.0), (4096, 4096, 512, 0.5), (65536, 512, 512, 1.0), (7936, 768, 192, 1.0), (263168, 512, 2048, 1.0)]:
    torch.manual_seed(0)
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    got = ops.gemm(a, b, M, N, K, alpha=alpha).float()
    want = alpha * (a.float() @ b.float().t())
    err, scale = float((got - want).abs().max()), float(want.abs().max())
    ok = err <= scale * 2.0 ** -7 and bool(torch.isfinite(got).all())
    bad += not ok
    print(f"check {M}x{N}x{K} alpha={alpha}: max err {err:.3e} scale {scale:.3e} {'ok' if ok else 'FAIL'}", flush=True)
def timeit(fn, iters=20, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
M = 263168
for (N, K) in [(1536, 512), (4096, 512), (512, 512), (512, 2048), (512, 4096), (1536, 512)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm(a, b, M, N, K, out=out))
    print(f"XCLIP_GEMM={os.environ.get('XCLIP_GEMM', '-')}  NT M={M} N={N} K={K}: {t * 1e3:8.1f} us  {2.0 * M * N * K / t / 1e9:7.1f} TF/s", flush=True)
print("failures", bad)
