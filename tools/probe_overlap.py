"""round 6: do an MFMA-bound GEMM and an HBM-bound row kernel of ANOTHER stream share the chip?  The step is the sum of its families (every
kernel takes all 256 CUs) although the GEMMs leave 80 % of the HBM bandwidth and the row kernels all of the matrix
pipe idle.  Each pair below: kernel A alone, kernel B alone, then both issued to two streams back to back, 20 rounds; 'together' near
max(A, B) = they overlap, near A + B = they queue.  (The ring GEMM holds all 160 KiB of a CU's LDS: a kernel that needs LDS cannot join it.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_clip_amd import ops
dev = torch.device("cuda:0")
M, D, F = 263168, 512, 2048
x = torch.randn(M, D, device=dev).bfloat16()
w1 = torch.randn(2 * F, D, device=dev).bfloat16() * 0.05
g = torch.ones(D, device=dev).bfloat16()
gi = torch.ones(F, device=dev).bfloat16()
u = ops.gemm(x, w1, M, 2 * F, D)
qkv = torch.randn(1024, 257, 3 * 512, device=dev).bfloat16()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(fa, fb, rounds=20):
    def timed(fns):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(rounds):
            for (st, f) in fns:
                with torch.cuda.stream(st):
                    f()
        for (st, _) in fns:
            torch.cuda.current_stream().wait_stream(st)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / rounds * 1e3
    for st in (s1, s2):
        st.wait_stream(torch.cuda.current_stream())
    for f in (fa, fb):
        for _ in range(3):
            f()
    return timed([(s1, fa)]), timed([(s2, fb)]), timed([(s1, fa), (s2, fb)])


def gemm_ff1():
    ops.gemm(x, w1, M, 2 * F, D, out=u)


def ln_plain():
    ops.layernorm_fwd(x, g)


def ln_geglu():
    ops.layernorm_fwd(u, gi, geglu=True)


def attn_fwd():
    ops.attention_fwd(qkv, None, 8, 0.125)


for name, fa, fb in [("FF1 GEMM (MFMA) + LayerNorm forward D = 512 (no LDS)", gemm_ff1, ln_plain),
                     ("FF1 GEMM (MFMA) + attention forward (78 KiB LDS per work-group)", gemm_ff1, attn_fwd),
                     ("LayerNorm forward + attention forward", ln_plain, attn_fwd),
                     ("FF1 GEMM + FF1 GEMM", gemm_ff1, gemm_ff1)]:
    a, b, t = run(fa, fb)
    print(f"{name}: A {a:7.1f} us   B {b:7.1f} us   together {t:7.1f} us   (sum {a + b:7.1f}, max {max(a, b):7.1f})", flush=True)
