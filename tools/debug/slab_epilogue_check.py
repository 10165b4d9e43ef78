"""which elements of a split-K (fp32 slab) GEMM differ from torch: per 256 x 256 tile and per 32-row / 32-column block inside the worst tile"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from x_clip_amd import ops
dev = torch.device("cuda:0")
for (M, N, K) in [(512, 512, 2048), (1000, 1536, 512), (520, 512, 2048)]:
    torch.manual_seed(0)
    a = torch.randn(K, M, device=dev, dtype=torch.bfloat16); b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    got = ops.gemm(a, b, M, N, K, True, True).float()
    want = a.float().t() @ b.float()
    err = (got - want).abs()
    print(f"tn {M}x{N}x{K}: max err {float(err.max()):.3e}")
    for m0 in range(0, M, 256):
        print("   ", " ".join(f"{float(err[m0:m0+256, n0:n0+256].max()):9.2e}" for n0 in range(0, N, 256)))
    bad = (err > 1.0).nonzero()
    if len(bad):
        r, c = bad[:, 0], bad[:, 1]
        print("    bad rows mod 256:", sorted(set((r % 256).tolist()))[:40], " bad cols mod 256:", sorted(set((c % 256).tolist()))[:70], "count", len(bad))
