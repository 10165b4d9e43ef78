"""random GEMM shapes / layouts / epilogue terms against torch on the GPU (hardware-only hazards do not show on the emulator):
    python tools/debug/gemm_fuzz.py [cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from x_clip_amd import ops
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = torch.Generator().manual_seed(seed)
def ri(lo, hi): return int(torch.randint(lo, hi + 1, (1,), generator=g))
bad = 0
for it in range(cases):
    lay = ["nt", "nn", "tn"][ri(0, 2)]
    M = 8 * ri(16, 200) if ri(0, 3) else 256 * ri(1, 6)
    N = 8 * ri(16, 200) if ri(0, 3) else 256 * ri(1, 6)
    K = 64 * ri(1, 40) if lay != "tn" else 64 * ri(4, 300)
    res = lay == "nt" and ri(0, 2) == 0
    alpha = [1.0, 0.5, 0.125][ri(0, 2)] if not res else 1.0
    ak, bk = lay[0] == "t", lay[1] == "n"
    a = torch.randn((K, M) if ak else (M, K), device=dev, dtype=torch.bfloat16)
    b = torch.randn((K, N) if bk else (N, K), device=dev, dtype=torch.bfloat16)
    r = torch.randn(M, N, device=dev, dtype=torch.bfloat16) if res else None
    got = ops.gemm(a, b, M, N, K, ak, bk, alpha=alpha, residual=r).float()
    want = alpha * ((a.float().t() if ak else a.float()) @ (b.float() if bk else b.float().t()))
    if res: want = want + r.float()
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    ok = err <= scale * 2.0 ** -7 and bool(torch.isfinite(got).all())     # 2 bf16 ulps of the output scale
    if not ok:
        bad += 1
        print(f"FAIL {lay} M={M} N={N} K={K} res={res} alpha={alpha}: err {err:.3e} scale {scale:.3e}", flush=True)
print(f"{cases} cases, {bad} failures")
