"""contrastive-head kernels at the BASELINE configs[2] per-rank block shape: b = 4096 local rows against B = 32768 gathered
columns, d = 512, bf16 (row block of the rank-sharded loss)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_clip_amd import _lib, ops
if "--measure" in sys.argv:          # libxclip_hip_measure.so: XCLIP_SIM=3 selects the round-1 loop (simloss3.h) for the A/B
    _lib.use_measurement_build()
dev = torch.device("cuda:0")
b, B, d = 4096, 32768, 512


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
    return s.elapsed_time(e) / iters * 1e-3


T = torch.nn.functional.normalize(torch.randn(b, d, device=dev), dim=-1).bfloat16()
I = torch.nn.functional.normalize(torch.randn(B, d, device=dev), dim=-1).bfloat16()
tau = torch.tensor([1.0], device=dev)
loss = torch.zeros(1, device=dev)
fl = 2.0 * b * B * d
t = timeit(lambda: ops.simloss_fwd(T, I, 1.0, 0, True, 1.0 / (2 * B), loss, log_scale=tau))
print(f"sim + online-LSE forward (logits never stored) [{b} x {B} x {d}]: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s   HBM algorithmic {(b+B)*d*2/t/1e9:7.1f} GB/s")
lse, _ = ops.simloss_fwd(T, I, 1.0, 0, True, 1.0 / (2 * B), loss, log_scale=tau)
lk = torch.full((B,), float(lse.mean()), device=dev)
dtau = torch.zeros(1, device=dev)
G = torch.empty(b, B, dtype=torch.bfloat16, device=dev)
t = timeit(lambda: ops.simloss_grad(T, I, 1.0, 0, True, 0.5 / B, 0.5 / B, 1.0 / B, lse, lk, dtau, log_scale=tau, times_scale=True, out=G))
print(f"sim gradient factor G (bf16 [{b} x {B}] written once): {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s   HBM write {b*B*2/t/1e9:7.1f} GB/s ({b*B*2/t/8e12*100:4.1f} % of 8 TB/s)")
G.zero_(); dtau.zero_()
ops.simloss_grad(T, I, 1.0, 0, True, 0.5 / B, 0.5 / B, 1.0 / B, lse, lk, dtau, log_scale=tau, times_scale=True, out=G)
print(f"   (check values: sum |G| = {float(G.float().abs().sum()):.6e}, G[0, 0] = {float(G[0, 0]):.4e}, G[4095, 32767] = {float(G[-1, -1]):.4e}, d tau = {float(dtau):.6e})")
if "--g-only" in sys.argv:
    sys.exit(0)
t = timeit(lambda: ops.gemm(G, I, b, d, B, b_kmajor=True))
print(f"dT = G I      (NN, K = {B}): {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s   reads G {b*B*2/t/1e9:7.1f} GB/s")
t = timeit(lambda: ops.gemm(G, T, B, d, b, a_kmajor=True, b_kmajor=True))
print(f"dI = G^T T    (TN, K = {b}): {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s   reads G {b*B*2/t/1e9:7.1f} GB/s")
out = torch.empty(b, B, dtype=torch.bfloat16, device=dev)
t = timeit(lambda: ops.gemm(T, I, b, B, d, out=out))
print(f"plain GEMM of the same shape (logits materialised, bf16): {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
for (bb, BB) in [(4096, 4096), (1024, 32768), (8192, 8192)]:
    T2, I2 = T[:bb].contiguous(), I[:BB].contiguous()
    t = timeit(lambda: ops.simloss_fwd(T2, I2, 1.0, 0, False, 1.0, loss, log_scale=tau)) if BB <= B and bb <= b else 0
    if t:
        print(f"sim fwd [{bb} x {BB}]: {t*1e6:8.1f} us  {2.0*bb*BB*d/t/1e12:7.1f} TF/s")
