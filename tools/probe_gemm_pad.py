"""Does the row stride of the operands matter (L2 channel spread)?  Times xclip_gemm on views with padded leading dimensions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_clip_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for (M, N, K) in [(263168, 1536, 512), (263168, 512, 2048)]:
    for pa, pb, pc in [(0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (64, 64, 64), (32, 32, 0), (8, 8, 0)]:
        a = torch.randn(M, K + pa, device=dev, dtype=torch.bfloat16)[:, :K]
        b = torch.randn(N, K + pb, device=dev, dtype=torch.bfloat16)[:, :K]
        out = torch.empty(M, N + pc, device=dev, dtype=torch.bfloat16)[:, :N]
        t = timeit(lambda: ops.gemm(a, b, M, N, K, out=out))
        print(f"M={M} N={N} K={K} pad A {pa:2d} B {pb:2d} C {pc:2d}: {t*1e3:8.1f} us  {2*M*N*K/t/1e9:7.1f} TF/s", flush=True)
