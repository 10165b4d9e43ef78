"""GPU micro-benchmarks of the hot kernels at the BASELINE cfg2 shapes (b=1024): prints achieved TFLOP/s / GB/s.
    python tools/probe_perf.py            (on the MI355X)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_clip_amd import ops

dev = torch.device("cuda:0")
bf = torch.bfloat16


def timeit(fn, iters=20, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def gemm_case(name, M, N, K, a_k, b_k, residual=False):
    a = torch.randn((K, M) if a_k else (M, K), device=dev, dtype=bf)
    b = torch.randn((K, N) if b_k else (N, K), device=dev, dtype=bf)
    out = torch.empty(M, N, device=dev, dtype=bf)
    res = torch.randn(M, N, device=dev, dtype=bf) if residual else None
    t = timeit(lambda: ops.gemm(a, b, M, N, K, a_k, b_k, residual=res, out=out))
    ref = ""
    if os.environ.get("XCLIP_PROBE_NO_REF") != "1":
        tr = timeit(lambda: torch.matmul(a.t() if a_k else a, b if b_k else b.t()))
        ref = f"   (hipBLASLt {tr*1e3:8.3f} ms {2*M*N*K/tr/1e12:7.1f} TF/s)"
    print(f"{name:28s} M={M:7d} N={N:5d} K={K:7d}  {t*1e3:8.3f} ms  {2*M*N*K/t/1e12:7.1f} TF/s{ref}", flush=True)


Mt = 1024 * 257
# the part needs ~50 ms of load before it settles at its sustained clock: the FIRST case of a process otherwise reads 10-15 % slow (the
# QKV forward measured 0.46 ms here for a long time and 0.40 ms as the fifth case of tools/probe_gemm_tail.py, same box, same kernel)
_a, _b = torch.randn(Mt, 512, device=dev, dtype=bf), torch.randn(1536, 512, device=dev, dtype=bf)
for _ in range(150):
    ops.gemm(_a, _b, Mt, 1536, 512)
torch.cuda.synchronize()
del _a, _b
gemm_case("qkv fwd (NT)", Mt, 1536, 512, False, False)
gemm_case("ff1 fwd (NT)", Mt, 4096, 512, False, False)
gemm_case("ff2 fwd (NT)", Mt, 512, 2048, False, False)
gemm_case("ff2 fwd + skip (NT)", Mt, 512, 2048, False, False, residual=True)
gemm_case("out fwd (NT)", Mt, 512, 512, False, False)
gemm_case("ff1 dgrad (NN)", Mt, 512, 4096, False, True)
gemm_case("ff2 dgrad (NN)", Mt, 2048, 512, False, True)
gemm_case("ff1 wgrad (TN)", 4096, 512, Mt, True, True)
gemm_case("ff2 wgrad (TN)", 512, 2048, Mt, True, True)
gemm_case("qkv wgrad (TN)", 1536, 512, Mt, True, True)
gemm_case("patch embed (NT)", 1024 * 32, 512, 3072, False, False)
gemm_case("out dgrad (NN)", Mt, 512, 512, False, True)
gemm_case("qkv dgrad (NN)", Mt, 512, 1536, False, True)
gemm_case("out wgrad (TN)", 512, 512, Mt, True, True)
if os.environ.get("XCLIP_PROBE_GEMM_ONLY") == "1":
    sys.exit(0)

b, n, h = 1024, 257, 8
qkv = torch.randn(b, n, 3 * h * 64, device=dev, dtype=bf)
mask = torch.ones(b, n, dtype=torch.bool, device=dev)
t = timeit(lambda: ops.attention_fwd(qkv, mask, h, 0.125))
fl = 4 * b * h * n * n * 64
print(f"attention fwd n=257        {t*1e3:8.3f} ms  {fl/t/1e12:7.1f} TF/s  ({(qkv.numel()*2*4/3)/t/1e9:7.0f} GB/s)")
out, lse = ops.attention_fwd(qkv, mask, h, 0.125)
do = torch.randn_like(out)
t = timeit(lambda: ops.attention_bwd(qkv, mask, out, do, lse, h, 0.125))
print(f"attention bwd n=257        {t*1e3:8.3f} ms  {2.5*fl/t/1e12:7.1f} TF/s (2.5x fwd flops)")

x = torch.randn(Mt, 512, device=dev, dtype=bf); g = torch.ones(512, device=dev, dtype=bf)
t = timeit(lambda: ops.layernorm_fwd(x, g))
print(f"layernorm fwd [{Mt},512]     {t*1e3:8.3f} ms  {2*x.numel()*2/t/1e9:7.0f} GB/s")
y, mean, rstd = ops.layernorm_fwd(x, g)
t = timeit(lambda: ops.layernorm_bwd(y, x, g, mean, rstd))
print(f"layernorm bwd [{Mt},512]     {t*1e3:8.3f} ms  {3*x.numel()*2/t/1e9:7.0f} GB/s")
uv = torch.randn(Mt, 4096, device=dev, dtype=bf); g2 = torch.ones(2048, device=dev, dtype=bf)
t = timeit(lambda: ops.layernorm_fwd(uv, g2, None, True))
print(f"geglu+ln fwd [{Mt},4096]    {t*1e3:8.3f} ms  {(uv.numel()*2*1.5)/t/1e9:7.0f} GB/s")
h2, mean, rstd = ops.layernorm_fwd(uv, g2, None, True)
t = timeit(lambda: ops.layernorm_bwd(h2, uv, g2, mean, rstd, True))
print(f"geglu+ln bwd [{Mt},4096]    {t*1e3:8.3f} ms  {(uv.numel()*2*2.5)/t/1e9:7.0f} GB/s")
