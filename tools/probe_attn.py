"""attention layout experiment: same work as (b=1024, n=257, heads=8) but as batch=8192 single-head problems (row stride
384 B instead of 3072 B) -- separates the HBM access-pattern cost from the kernel's own cost."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_clip_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
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


for (b, n, h) in [(1024, 257, 8), (8192, 257, 1), (1024, 256, 8), (1024, 288, 8), (1024, 32, 8), (8192, 32, 1)]:
    qkv = torch.randn(b, n, 3 * h * 64, device=dev, dtype=torch.bfloat16)
    mask = torch.ones(b, n, dtype=torch.bool, device=dev)
    t = timeit(lambda: ops.attention_fwd(qkv, mask, h, 0.125))
    out, lse = ops.attention_fwd(qkv, mask, h, 0.125)
    do = torch.randn_like(out)
    tb = timeit(lambda: ops.attention_bwd(qkv, mask, out, do, lse, h, 0.125))
    gb = qkv.numel() * 2 * (4 / 3) / 1e9
    print(f"b={b:5d} n={n:4d} h={h}: fwd {t*1e3:8.1f} us ({gb/t*1e3:6.0f} GB/s)   bwd {tb*1e3:8.1f} us ({gb*2/tb*1e3:6.0f} GB/s)", flush=True)
# reference: plain copy of the same bytes
x = torch.randn(1024 * 257 * 1536, device=dev, dtype=torch.bfloat16)
y = torch.empty_like(x)
t = timeit(lambda: y.copy_(x))
print(f"copy of qkv-size buffer: {t*1e3:8.1f} us ({x.numel()*4/t/1e6:6.0f} GB/s)")
