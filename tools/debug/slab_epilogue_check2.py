"""where do the values of a wrong split-K tile come from?  K = one slice only would be cleaner, so use splits via a small K and compare
against the per-slice partial sums: got = sum over slices; here we just look for each got element's position in want"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from x_clip_amd import ops
dev = torch.device("cuda:0")
M, N, K = 256, 256, 4096
torch.manual_seed(0)
a = torch.randn(K, M, device=dev, dtype=torch.bfloat16); b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
got = ops.gemm(a, b, M, N, K, True, True).float().cpu()
want = (a.float().t() @ b.float()).cpu()
print("max err", float((got - want).abs().max()))
wf = want.flatten()
for (r, c) in [(0, 0), (0, 1), (0, 4), (0, 8), (0, 32), (0, 64), (1, 0), (8, 0), (16, 0), (31, 0), (32, 0), (33, 5), (128, 0), (200, 77)]:
    d = (wf - got[r, c]).abs()
    j = int(d.argmin())
    print(f"got[{r:3d},{c:3d}] = {float(got[r, c]):9.3f}  want there {float(want[r, c]):9.3f}  nearest want at ({j // N:3d},{j % N:3d}) diff {float(d[j]):.3f}")
