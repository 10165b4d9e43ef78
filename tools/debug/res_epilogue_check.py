"""where does C = A B + R differ from the reference? (debug aid for the residual epilogue of gemm4.h)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from x_clip_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
zero_res = False
for (M, N, K) in [(256, 256, 64), (512, 256, 64), (4096, 512, 64), (4096, 512, 2048), (65536, 512, 2048), (263168, 512, 2048)]:
    a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16()
    r = torch.randn(M, N, device=dev).bfloat16()
    c = ops.gemm(a, b, M, N, K, residual=r)
    ref = (a.float() @ b.float().t() + (0 if zero_res else r.float()))
    plain = ops.gemm(a, b, M, N, K).float()
    err = (c.float() - ref).abs()
    bad = (err > 0.1 * ref.abs().max()).nonzero()
    print(f"M={M} N={N} K={K}: max err {float(err.max()):.3f} (scale {float(ref.abs().max()):.1f}), bad elements {bad.shape[0]}, NaN {int(torch.isnan(c.float()).sum())}")
    if bad.shape[0]:
        rows = sorted(set(bad[:, 0].tolist()))[:12]; cols = sorted(set(bad[:, 1].tolist()))[:24]
        print("  first bad rows", rows, "cols", cols)
        i, j = bad[0].tolist()
        print(f"  at ({i},{j}): got {float(c[i,j]):.3f} want {float(ref[i,j]):.3f} plain {float(plain[i,j]):.3f} res {float(r[i,j]):.3f}; got-plain {float(c[i,j])-float(plain[i,j]):.3f}")
        import collections
        cnt = collections.Counter()
        for i, j in bad.tolist()[:4000]:
            ti, tj = i % 256, j % 256
            cnt[(f"rowblk {(ti % 128) // 32} wm {ti // 128}", f"lane {ti % 32:2d}", f"wn {tj // 64} j {(tj % 64) // 32} col-in-32 {tj % 32:2d}")] += 1
        for k, v in sorted(cnt.items())[:40]:
            print("   ", k, v)
        c2 = ops.gemm(a, b, M, N, K, residual=r)
        bad2 = ((c2.float() - ref).abs() > 0.1 * ref.abs().max()).nonzero()
        print("  second launch: bad", bad2.shape[0], "same positions:", bool(bad2.shape == bad.shape and torch.equal(bad2, bad)))
