"""Time the LayerNorm kernels at the bench shapes (MI355X).  usage: python tools/probe_ln.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_clip_amd import _lib, ops
if "--measure" in sys.argv:          # libxclip_hip_measure.so: XCLIP_LN_FWD = 0 (one row per wave) / 1 / 2 / 4 rows per wave for the narrow-row forward
    _lib.use_measurement_build()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda")
    for rows, dim, geglu in [(1024 * 257, 2048, True), (1024 * 33, 2048, True), (1024 * 257, 512, False), (1024 * 33, 512, False), (2 * 2048 * 289, 1024, False)]:
        w = 2 * dim if geglu else dim
        x = torch.randn(rows, w, device=dev, dtype=torch.bfloat16)
        g = torch.ones(dim, device=dev, dtype=torch.bfloat16)
        dy = torch.randn(rows, dim, device=dev, dtype=torch.bfloat16)
        y, mean, rstd = ops.layernorm_fwd(x, g, None, geglu)
        tf = timeit(lambda: ops.layernorm_fwd(x, g, None, geglu))
        tb = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, geglu))
        bf = rows * (w + dim) * 2
        bb = rows * (w + dim + w) * 2
        print(f"rows={rows} dim={dim} geglu={geglu}: fwd {tf:8.1f} us ({bf / tf / 1e3:6.0f} GB/s)   bwd {tb:8.1f} us ({bb / tb / 1e3:6.0f} GB/s)")
        if not geglu:
            r = torch.randn(rows, dim, device=dev, dtype=torch.bfloat16)
            tr = timeit(lambda: ops.layernorm_fwd(x, g, r, False))
            print(f"rows={rows} dim={dim} + residual: fwd {tr:8.1f} us ({rows * 3 * dim * 2 / tr / 1e3:6.0f} GB/s)")


if __name__ == "__main__":
    main()
