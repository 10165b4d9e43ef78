"""tile-quantisation tail of the persistent GEMM: M = 1024 x 257 rows is 1028 row tiles, so the N = 512 products have 2056 tiles = 8.03
rounds of 256 work-groups.  python tools/probe_gemm_tail.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_clip_amd import ops
dev = torch.device("cuda")


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for N, K in ((512, 512), (512, 2048), (512, 1536), (512, 4096), (1536, 512), (4096, 512)):
    for M in (1024 * 256, 1024 * 257):
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm(a, w, M, N, K))
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        print(f"NT M={M} N={N} K={K}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TF/s   tiles {tiles} = {tiles / 256:.2f} rounds")
