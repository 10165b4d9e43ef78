"""random GEMM shapes / layouts / epilogue terms against torch on the GPU (hardware-only hazards do not show on the emulator):
    python tools/debug/gemm_fuzz.py [cases] [seed] [--large | --small]
--small: shapes gemm_small.h takes (M, N, K multiples of 64, at most 32 K steps, a few tiles of output; any layout, alpha, the skip term with any
layout, in place): counted waits on a ring of 2 ... 8 LDS-DMA stages, which the emulator cannot see
--large: shapes of more than one round of tiles (the row-tail cut of persistent launches, xclip_api.hip gemm2_tail_cut: 257 ... 1100 tiles, ragged
last row tile, with / without the skip term, in place) and weight-gradient shapes with long contractions (4 ... 48 output tiles x up to 64 K slices on
the 1-D split-K grid, gemm2.h g2_where)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from x_clip_amd import ops
dev = torch.device("cuda:0")
large = "--large" in sys.argv
small = "--small" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
cases = int(argv[0]) if len(argv) > 0 else 300
seed = int(argv[1]) if len(argv) > 1 else 0
g = torch.Generator().manual_seed(seed)
def ri(lo, hi): return int(torch.randint(lo, hi + 1, (1,), generator=g))
bad = 0
for it in range(cases):
    lay = ["nt", "nn", "tn"][ri(0, 2)]
    M = 8 * ri(16, 200) if ri(0, 3) else 256 * ri(1, 6)
    N = 8 * ri(16, 200) if ri(0, 3) else 256 * ri(1, 6)
    K = 64 * ri(1, 40) if lay != "tn" else 64 * ri(4, 300)
    if large and lay != "tn":
        N = 256 * ri(1, 6) if ri(0, 2) else 8 * ri(40, 190)
        tiles_n = (N + 255) // 256
        M = 256 * ri(max(1, 257 // tiles_n), 1100 // tiles_n) + (8 * ri(0, 31) if ri(0, 1) else 0)
        K = 64 * ri(2, 48)
    elif large:
        M, N = 256 * ri(1, 8) if ri(0, 1) else 8 * ri(32, 250), 256 * ri(1, 6) if ri(0, 1) else 8 * ri(32, 190)
        K = 64 * ri(300, 4200)
    if small:
        M, N, K = 64 * ri(1, 24), 64 * ri(1, 24), 64 * ri(1, 32)
        if ri(0, 5) == 0:
            M, N = 64 * ri(17, 40), 64 * ri(17, 40)                    # more than 256 tiles: four stages, two work-groups per CU
            K = 64 * ri(1, 8)
    res = (lay == "nt" or small) and ri(0, 2) == 0
    alpha = [1.0, 0.5, 0.125][ri(0, 2)] if not res else 1.0
    ak, bk = lay[0] == "t", lay[1] == "n"
    a = torch.randn((K, M) if ak else (M, K), device=dev, dtype=torch.bfloat16)
    b = torch.randn((K, N) if bk else (N, K), device=dev, dtype=torch.bfloat16)
    r = torch.randn(M, N, device=dev, dtype=torch.bfloat16) if res else None
    inplace = res and ri(0, 1) == 0
    out = r.clone() if inplace else None
    got = ops.gemm(a, b, M, N, K, ak, bk, alpha=alpha, residual=(out if inplace else r), out=out).float()
    want = alpha * ((a.float().t() if ak else a.float()) @ (b.float() if bk else b.float().t()))
    if res: want = want + r.float()
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    tol = 2.0 ** -7                                                       # 2 bf16 ulps of the output scale
    ok = err <= scale * tol and bool(torch.isfinite(got).all())
    if not ok:
        bad += 1
        print(f"FAIL {lay} M={M} N={N} K={K} res={res} alpha={alpha}: err {err:.3e} scale {scale:.3e}", flush=True)
if small:
    assert ops.gemm_small_limit() > 0
print(f"{cases} cases, {bad} failures" + (" (gemm_small.h shapes)" if small else ""))
