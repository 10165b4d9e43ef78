import os, sys
sys.path.insert(0, os.getcwd())
import torch
from x_clip_amd import ops
dev = torch.device("cuda")
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
rows, dim = 1024 * 257, 512
p = torch.randn(rows, dim, device=dev, dtype=torch.bfloat16); x = torch.randn_like(p); g1 = torch.ones(dim, device=dev, dtype=torch.bfloat16); g2 = g1.clone()
dh2 = torch.randn_like(p); dx2 = torch.randn_like(p)
def sep_f():
    x1, m2, r2 = ops.layernorm_fwd(p, g1, res=x); h2, m3, r3 = ops.layernorm_fwd(x1, g2)
x1, m2, r2, h2, m3, r3 = ops.layernorm_chain_fwd(p, g1, x, g2)
dg = torch.zeros(2, dim, device=dev, dtype=torch.float32)
def sep_b():
    dx1, _ = ops.layernorm_bwd(dh2, x1, g2, m3, r3, dres=dx2, dg=dg[0]); dp, _ = ops.layernorm_bwd(dx1, p, g1, m2, r2, dg=dg[1])
print(f"fwd separate {timeit(sep_f):.1f} us  chain {timeit(lambda: ops.layernorm_chain_fwd(p, g1, x, g2)):.1f} us")
print(f"bwd separate {timeit(sep_b):.1f} us  chain {timeit(lambda: ops.layernorm_chain_bwd(dh2, x1, g2, m3, r3, dx2, p, g1, m2, r2, dg[0], dg[1])):.1f} us")
