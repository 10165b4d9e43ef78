"""runs ONE GEMM shape a few times (for rocprofv3 --pmc): python tools/probe_gemm_one.py M N K layout [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_clip_amd import ops
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
layout = sys.argv[4] if len(sys.argv) > 4 else "nt"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
a_k, b_k = layout == "tn", layout in ("nn", "tn")
dev = torch.device("cuda:0")
a = torch.randn((K, M) if a_k else (M, K), device=dev, dtype=torch.bfloat16)
b = torch.randn((K, N) if b_k else (N, K), device=dev, dtype=torch.bfloat16)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(iters):
    ops.gemm(a, b, M, N, K, a_k, b_k, out=out)
torch.cuda.synchronize()
