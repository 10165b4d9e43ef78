"""gemm8.h (the hand-scheduled asm ring GEMM) against gemm4.h's g5_run, same process, measurement build:
    python tools/probe_gemm8.py check            bit-exact comparison on small / medium / banded shapes, three runs each (race screen)
    python tools/probe_gemm8.py time [quick]     interleaved A/B timing on the shapes of BASELINE configs[1] (b = 1024) that gemm8 takes
The two kernels run the same MFMA sequence per output element (same K order, fp32 accumulators, one rounding): any difference is a bug."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_clip_amd import _lib, ops
_lib.use_measurement_build()
L = _lib.lib()
import ctypes
L.xclip_measure_gemm8.restype = ctypes.c_int
L.xclip_measure_gemm8.argtypes = [ctypes.c_int]
dev = torch.device("cuda:0")
bf = torch.bfloat16


def run(a, b, M, N, K, bk, on, out=None):
    """on: False / 0 = g5_run, True / 1 = gemm8 as the product would pick it, 2.. = a measurement variant (gemm8_gen.py VARIANTS), 10 / 11 = variant 0 / 1"""
    L.xclip_measure_gemm8(int(on))
    r = ops.gemm(a, b, M, N, K, False, bk, out=out)
    L.xclip_measure_gemm8(0)
    return r


def check():
    bad = 0
    shapes = [(2048, 512, 512), (2048, 512, 1024), (8192, 1536, 512), (4096, 2048, 576), (65536, 1536, 512), (16384, 4096, 512), (32768, 512, 2048),
              (262144, 512, 512), (65536, 512, 4096), (24576, 768, 1536)]
    for (M, N, K) in shapes:
        for bk in (False, True):
            g = torch.Generator(device="cpu").manual_seed(M + N + K)
            a = (torch.randn(M, K, generator=g) * 0.5).to(dev, bf)
            b = (torch.randn((K, N) if bk else (N, K), generator=g) * 0.5).to(dev, bf)
            ref = run(a, b, M, N, K, bk, False)
            torch.cuda.synchronize()
            worst = 0
            for rep in range(3):
                out = torch.full((M, N), float("nan"), device=dev, dtype=bf)
                run(a, b, M, N, K, bk, True, out=out)
                torch.cuda.synchronize()
                neq = (out.view(torch.int16) != ref.view(torch.int16))
                n = int(neq.sum())
                worst = max(worst, n)
                if n:
                    idx = neq.nonzero()
                    rows = idx[:, 0].unique()
                    cols = idx[:, 1].unique()
                    d = (out.float() - ref.float())
                    nan = int(torch.isnan(out.float()).sum())
                    print(f"   rep {rep}: {n} elements differ ({nan} NaN = never stored); rows {rows[:6].tolist()}..{rows[-3:].tolist()} ({len(rows)}), "
                          f"cols {cols[:6].tolist()}..{cols[-3:].tolist()} ({len(cols)}); max |d| {float(torch.nan_to_num(d).abs().max()):.4g}; "
                          f"first {idx[0].tolist()} out {float(out[idx[0][0], idx[0][1]]):.5g} ref {float(ref[idx[0][0], idx[0][1]]):.5g}", flush=True)
            # (the fp32 reference: guards against both kernels sharing a bug)
            sel = torch.randint(0, M, (64,), device=dev)
            want = a[sel].float() @ (b.float() if bk else b.float().t())
            err = float((ref[sel].float() - want).abs().max() / want.abs().max())
            print(f"gemm8 check M={M:6d} N={N:5d} K={K:5d} {'NN' if bk else 'NT'}: {'EXACT' if worst == 0 else 'MISMATCH ' + str(worst)}   (g5 vs fp32 rows: {err:.2e})", flush=True)
            bad += worst != 0
    print("gemm8 check:", "ALL EXACT" if bad == 0 else f"{bad} shapes differ")
    return bad


def timeit_pair(f0, f1, iters=20, warm=6, rounds=4):
    for _ in range(warm):
        f0(); f1()
    torch.cuda.synchronize()
    t = [[], []]
    for _ in range(rounds):
        for k, f in ((0, f0), (1, f1)):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                f()
            e.record()
            torch.cuda.synchronize()
            t[k].append(s.elapsed_time(e) / iters)
    return sorted(t[0])[len(t[0]) // 2], sorted(t[1])[len(t[1]) // 2]


def time_shapes(quick, variants):
    Mt, Mv = 1024 * 257, 1024 * 33
    shapes = [("ff1 fwd text", Mt, 4096, 512, False), ("qkv fwd text", Mt, 1536, 512, False), ("out fwd text", Mt, 512, 512, False),
              ("ff2 dgrad text", Mt, 2048, 512, True), ("out dgrad text", Mt, 512, 512, True), ("ff1 dgrad text", Mt, 512, 4096, True),
              ("qkv dgrad text", Mt, 512, 1536, True), ("ff1 fwd vision", Mv, 4096, 512, False), ("qkv fwd vision", Mv, 1536, 512, False),
              ("ff2 dgrad vision", Mv, 2048, 512, True), ("ff1 dgrad vision", Mv, 512, 4096, True)]
    if quick:
        shapes = [shapes[1], shapes[0], shapes[3], shapes[5]]
    a0 = torch.randn(Mt, 512, device=dev, dtype=bf)
    b0 = torch.randn(1536, 512, device=dev, dtype=bf)
    for _ in range(100):
        ops.gemm(a0, b0, Mt, 1536, 512)
    torch.cuda.synchronize()
    tot = {}
    for (name, M, N, K, bk) in shapes:
        a = torch.randn(M, K, device=dev, dtype=bf)
        b = torch.randn((K, N) if bk else (N, K), device=dev, dtype=bf)
        o0 = torch.empty(M, N, device=dev, dtype=bf)
        o1 = torch.empty(M, N, device=dev, dtype=bf)
        fl = 2.0 * M * N * K
        line = f"{name:18s} M={M:6d} N={N:5d} K={K:5d} {'NN' if bk else 'NT'}:"
        for var in variants:
            t0, t1 = timeit_pair(lambda: run(a, b, M, N, K, bk, 0, out=o0), lambda: run(a, b, M, N, K, bk, var, out=o1), iters=10, rounds=3)
            same = bool((o0.view(torch.int16) == o1.view(torch.int16)).all())
            tot[0] = tot.get(0, 0.0) + t0 / len(variants)
            tot[var] = tot.get(var, 0.0) + t1
            if var == variants[0]:
                line += f" g5 {t0*1e3:7.1f} us {fl/t0/1e9:6.0f} TF/s |"
            line += f" v{var} {t1*1e3:7.1f} ({100*(t0/t1-1):+5.1f} %{'' if same else ' x'}) |"
        print(line, flush=True)
    print("sum: " + ", ".join(f"{'g5' if k == 0 else 'v' + str(k)} {t*1e3:.1f} us" for k, t in tot.items()))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "check"
    if what == "check":
        sys.exit(1 if check() else 0)
    variants = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1]
    time_shapes(len(sys.argv) > 2 and sys.argv[2] == "quick", variants)
