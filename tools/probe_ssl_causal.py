"""BatchNorm / negative-cosine kernels of the SimSiam head and causal vs. full attention at the text-tower shape, on MI355X:
python tools/probe_ssl_causal.py"""
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


print("# SimSiam head kernels, bf16 (rows = 1024 images x 33 tokens, the configs[1] vision tower with patch dropout 0.5)")
for rows, C in ((1024 * 33, 4096), (1024 * 33, 256), (1024 * 65, 4096)):
    x = torch.randn(rows, C, device=dev, dtype=torch.bfloat16)
    dy = torch.randn_like(x)
    g = torch.ones(C, device=dev)
    b = torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    y, mean, rstd = ops.batchnorm_fwd(x, g, b, rm, rv, 0.1, 1e-5, True, True)
    tf = timeit(lambda: ops.batchnorm_fwd(x, g, b, rm, rv, 0.1, 1e-5, True, True))
    tb = timeit(lambda: ops.batchnorm_bwd(x, dy, g, b, mean, rstd, True, True, True))
    e = 2
    print(f"batchnorm+relu [{rows} x {C}]: fwd {tf:8.1f} us  {3 * rows * C * e / tf / 1e3:7.1f} GB/s (x read twice + y written)   "
          f"bwd {tb:8.1f} us  {5 * rows * C * e / tb / 1e3:7.1f} GB/s (x, dy read twice + dx written)")
rows, D = 1024 * 33, 256
p, z = torch.randn(rows, D, device=dev, dtype=torch.bfloat16), torch.randn(rows, D, device=dev, dtype=torch.bfloat16)
acc = torch.zeros(1, device=dev)
st = ops.neg_cosine_fwd(p, z, 1.0 / rows, acc)
gm = torch.ones(1, device=dev)
tf = timeit(lambda: ops.neg_cosine_fwd(p, z, 1.0 / rows, acc))
tb = timeit(lambda: ops.neg_cosine_bwd(p, z, st, gm, 1.0 / rows))
print(f"neg-cosine [{rows} x {D}]: fwd {tf:.1f} us  {2 * rows * D * 2 / tf / 1e3:.1f} GB/s   bwd {tb:.1f} us  {3 * rows * D * 2 / tb / 1e3:.1f} GB/s")

print("# causal vs. full attention, bf16, batch 1024 x 8 heads (the text tower's shape; the causal encoder has n = 256, no CLS token)")
for n in (256, 257):
    qkv = torch.randn(1024, n, 3 * 8 * 64, device=dev, dtype=torch.bfloat16)
    for causal in (False, True):
        out, lse = ops.attention_fwd(qkv, None, 8, 0.125, causal)
        do = torch.randn_like(out)
        tf = timeit(lambda: ops.attention_fwd(qkv, None, 8, 0.125, causal))
        tb = timeit(lambda: ops.attention_bwd(qkv, None, out, do, lse, 8, 0.125, causal))
        print(f"n = {n} causal = {int(causal)}: fwd {tf:7.1f} us   bwd {tb:7.1f} us")
