"""gemm_small.h against the 256 x 256 kernels on the small-output products of BASELINE configs[1] (b = 1024: the pooled last layer's
B-row products, the latent projections, their weight gradients), same process, interleaved (xclip_gemm_small_limit on / off):
    python tools/probe_gemm_small.py
prints per shape: microseconds and TF/s of both routes (median of 5 rounds of 20 launches each)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x_clip_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")
SHAPES = [("nt", 1024, 512, 512, False), ("nn", 1024, 512, 512, False), ("tn", 512, 512, 1024, False), ("nt", 1024, 512, 2048, True),
          ("tn", 4096, 512, 1024, False), ("nt", 1024, 4096, 512, False), ("nn", 1024, 512, 4096, False), ("tn", 512, 2048, 1024, False),
          ("tn", 1024, 512, 1024, False), ("nn", 1024, 2048, 512, False), ("nn", 1024, 512, 1024, False),
          # where the policy's bounds sit: more work behind the same few tiles
          ("nn", 2048, 512, 4096, False), ("tn", 512, 512, 8192, False), ("nt", 4096, 4096, 512, False), ("nt", 2048, 2048, 512, False)]


def run(layout, M, N, K, res, iters=20, rounds=5):
    a_k, b_k = layout == "tn", layout in ("nn", "tn")
    a = torch.randn((K, M) if a_k else (M, K), device=DEV).to(torch.bfloat16)
    b = torch.randn((K, N) if b_k else (N, K), device=DEV).to(torch.bfloat16)
    r = torch.randn(M, N, device=DEV).to(torch.bfloat16) if res else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    t = {"small": [], "big": []}
    default = ops.gemm_small_limit()
    for _ in range(rounds):
        for name, lim in (("small", 1 << 62), ("big", 0)):
            ops.gemm_small_limit(lim)
            for _ in range(3):
                ops.gemm(a, b, M, N, K, a_k, b_k, residual=r, out=out)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                ops.gemm(a, b, M, N, K, a_k, b_k, residual=r, out=out)
            e.record()
            torch.cuda.synchronize()
            t[name].append(s.elapsed_time(e) / iters * 1e3)
    ops.gemm_small_limit(default)
    us = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    fl = 2.0 * M * N * K
    takes = "default: small" if fl <= default and ((M + 255) // 256) * ((N + 255) // 256) <= 64 else "default: big"
    print(f"gemm {layout.upper()} M={M:5d} N={N:5d} K={K:5d}{' +res' if res else '     '}  small {us['small']:7.1f} us {fl / us['small'] * 1e-6:7.1f} TF/s"
          f"   big {us['big']:7.1f} us {fl / us['big'] * 1e-6:7.1f} TF/s   {takes}", flush=True)


if __name__ == "__main__":
    for sh in SHAPES:
        run(*sh)
