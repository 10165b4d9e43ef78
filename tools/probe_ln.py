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


def geglu_ab():
    """GEGLU-LayerNorm backward: grid size (XCLIP_LNG_BLOCKS), non-temporal accesses (XCLIP_ROWS_NT bit 1), round-3 LDS request (XCLIP_LNG_LDSPAD)."""
    dev = torch.device("cuda")
    for rows in (1024 * 257, 1024 * 33):
        dim = 2048
        x = torch.randn(rows, 2 * dim, device=dev, dtype=torch.bfloat16)
        g = torch.ones(dim, device=dev, dtype=torch.bfloat16)
        dy = torch.randn(rows, dim, device=dev, dtype=torch.bfloat16)
        _, mean, rstd = ops.layernorm_fwd(x, g, None, True)
        bb = rows * 5 * dim * 2
        for tag, env in [("round 3: 4096 groups, 36.9 KB LDS", dict(XCLIP_LNG_BLOCKS="4096", XCLIP_LNG_LDSPAD="1")),
                         ("1024", dict(XCLIP_LNG_BLOCKS="1024")), ("1536", dict(XCLIP_LNG_BLOCKS="1536")),
                         ("2048", dict(XCLIP_LNG_BLOCKS="2048")), ("3072", dict(XCLIP_LNG_BLOCKS="3072")),
                         ("4096", dict(XCLIP_LNG_BLOCKS="4096")),
                         ("1536 nt", dict(XCLIP_LNG_BLOCKS="1536", XCLIP_ROWS_NT="47")), ("3072 nt", dict(XCLIP_LNG_BLOCKS="3072", XCLIP_ROWS_NT="47"))]:
            for k in ("XCLIP_LNG_BLOCKS", "XCLIP_ROWS_NT", "XCLIP_LNG_LDSPAD"):
                os.environ.pop(k, None)
            os.environ.setdefault("XCLIP_ROWS_NT", "46")        # the non-temporal hint off unless the case asks for it
            os.environ.update(env)
            tb = min(timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, True)) for _ in range(3))
            print(f"rows={rows} geglu bwd [{tag:36s}]: {tb:8.1f} us ({bb / tb / 1e3:6.0f} GB/s)", flush=True)


def after_gemm():
    """Why the row kernels are slower inside the step than alone: the D = 512 LayerNorm forward (263,168 rows) timed alone, behind a GEMM that
    writes an UNRELATED buffer (power state / dirty lines in general), and behind the GEMM that writes ITS input (the step's situation)."""
    dev = torch.device("cuda")
    rows, dim = 1024 * 257, 512
    g = torch.ones(dim, device=dev, dtype=torch.bfloat16)
    x = torch.randn(rows, dim, device=dev, dtype=torch.bfloat16)
    z = torch.empty(rows, dim, device=dev, dtype=torch.bfloat16)
    a512 = torch.randn(rows, 512, device=dev, dtype=torch.bfloat16) * 0.05
    a4096 = torch.randn(rows, 4096, device=dev, dtype=torch.bfloat16) * 0.02
    w512 = torch.randn(512, 512, device=dev, dtype=torch.bfloat16)
    w4096 = torch.randn(512, 4096, device=dev, dtype=torch.bfloat16)

    def ln_time(before, n=12):
        ts = []
        for i in range(n + 3):
            before()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.layernorm_fwd(x, g, None, False)
            e1.record()
            torch.cuda.synchronize()
            if i >= 3:
                ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2], ts[0]

    cases = [("alone (idle chip in front)", lambda: None),
             ("behind GEMM K=512 -> unrelated buffer", lambda: ops.gemm(a512, w512, rows, 512, 512, out=z)),
             ("behind GEMM K=512 -> its input", lambda: ops.gemm(a512, w512, rows, 512, 512, out=x)),
             ("behind GEMM K=4096 -> unrelated buffer", lambda: ops.gemm(a4096, w4096, rows, 512, 4096, out=z)),
             ("behind GEMM K=4096 -> its input", lambda: ops.gemm(a4096, w4096, rows, 512, 4096, out=x)),
             ("behind another LayerNorm (unrelated)", lambda: ops.layernorm_fwd(a512, g, None, False)),
             ("behind 50 back-to-back LayerNorms", lambda: [ops.layernorm_fwd(a512, g, None, False) for _ in range(50)])]
    for tag, before in cases:
        med, best = ln_time(before)
        print(f"ln_fwd 263168 x 512 [{tag:42s}]: median {med:7.1f} us  best {best:7.1f} us  ({rows * 2 * dim * 2 / med / 1e3:5.0f} GB/s)", flush=True)


def main():
    if "--geglu-ab" in sys.argv:
        return geglu_ab()
    if "--after-gemm" in sys.argv:
        return after_gemm()
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
