"""Round-4 A/B probes on one box, measurement build (the switches are read once per process: run once per setting):
    python tools/probe_r4_ab.py attn      XCLIP_ATTN_ABL=4 -> the 257th token as a 33rd block (round-3 form); unset -> as the accumulators' initial value
    python tools/probe_r4_ab.py gemm      XCLIP_GEMM_TAIL=0 -> uncut persistent launches; unset -> the row tail as a split-K problem
    python tools/probe_r4_ab.py wgrad     XCLIP_GEMM_SPLIT2D=1 -> split-K weight gradients on the (tile, slice) grid; unset -> 1-D grid, an XCD holds whole K slices
    python tools/probe_r4_ab.py scatter   XCLIP_SCATTER_STAGED=0 -> the token-embedding gradient flushes a run with strided atomics; unset -> through the wave's LDS row
    python tools/probe_r4_ab.py wide      heads of 80 / 96 / 128 features (128-wide head slots, the tiled kernels of attention.h) beside the 64-wide head-resident ones
Prints one line per shape; the text / vision shapes of BASELINE configs[1] (b = 1024)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_clip_amd import _lib, ops
_lib.use_measurement_build()
dev = torch.device("cuda:0")
bf = torch.bfloat16


def timeit(fn, iters=20, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):                                   # three rounds: median of the round means
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters)
    return sorted(ts)[1]


def warm_clock():
    a, b = torch.randn(263168, 512, device=dev, dtype=bf), torch.randn(1536, 512, device=dev, dtype=bf)
    for _ in range(150):
        ops.gemm(a, b, 263168, 1536, 512)
    torch.cuda.synchronize()


tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("XCLIP_")) or "default"
what = sys.argv[1] if len(sys.argv) > 1 else "attn"
warm_clock()
if what == "attn":
    for (b, n, h, masked) in [(1024, 257, 8, True), (1024, 257, 8, False), (1024, 256, 8, True), (1024, 33, 8, False), (1024, 129, 8, True)]:
        qkv = torch.randn(b, n, 3 * h * 64, device=dev, dtype=bf)
        mask = torch.ones(b, n, dtype=torch.bool, device=dev) if masked else None
        t = timeit(lambda: ops.attention_fwd(qkv, mask, h, 0.125))
        out, lse = ops.attention_fwd(qkv, mask, h, 0.125)
        do = torch.randn_like(out)
        tb = timeit(lambda: ops.attention_bwd(qkv, mask, out, do, lse, h, 0.125))
        fl = 4.0 * b * h * n * n * 64
        print(f"[{tag}] attention b={b} n={n} h={h} mask={int(masked)}: fwd {t*1e3:7.1f} us ({fl/t/1e9:6.1f} TF/s)   bwd {tb*1e3:7.1f} us ({2*fl/tb/1e9:6.1f} TF/s algorithmic)", flush=True)
elif what == "scatter":
    for (b, n, vocab, zipf) in [(1024, 256, 10000, False), (1024, 256, 49408, False), (1024, 256, 49408, True)]:
        g = torch.Generator(device="cpu").manual_seed(0)
        if zipf:                                                 # frequencies ~ 1 / rank, like natural text
            w = 1.0 / torch.arange(1, vocab + 1, dtype=torch.float64)
            tok = torch.multinomial(w / w.sum(), b * n, replacement=True, generator=g).view(b, n)
        else:
            tok = torch.randint(0, vocab, (b, n), generator=g)
        tok = tok.to(dev)
        st = torch.sort(tok.flatten())
        dout = torch.randn(b * (n + 1), 512, device=dev, dtype=bf)
        dE = torch.zeros(vocab, 512, dtype=torch.float32, device=dev)
        t = timeit(lambda: ops.scatter_add_sorted(dout, st.values, st.indices, dE, n_in=n, n_out=n + 1, row_off=1))
        print(f"[{tag}] scatter_add_sorted {b} x {n} tokens, vocabulary {vocab}{' (zipf)' if zipf else ' (uniform)'}: {t*1e3:7.1f} us ({b * n * 1024 / t / 1e6:6.0f} GB/s of rows)", flush=True)
elif what == "wgrad":
    Mt, Mv = 1024 * 257, 1024 * 33
    for (name, M, N, K) in [("ff1 wgrad text", 4096, 512, Mt), ("ff2 wgrad text", 512, 2048, Mt), ("qkv wgrad text", 1536, 512, Mt), ("out wgrad text", 512, 512, Mt),
                            ("ff1 wgrad vision", 4096, 512, Mv), ("ff2 wgrad vision", 512, 2048, Mv), ("qkv wgrad vision", 1536, 512, Mv), ("out wgrad vision", 512, 512, Mv),
                            ("patch embed wgrad", 512, 3072, 32768)]:
        a = torch.randn(K, M, device=dev, dtype=bf)
        b = torch.randn(K, N, device=dev, dtype=bf)
        out = torch.empty(M, N, device=dev, dtype=bf)
        t = timeit(lambda: ops.gemm(a, b, M, N, K, a_kmajor=True, b_kmajor=True, out=out))
        print(f"[{tag}] {name:20s} M={M:5d} N={N:5d} K={K:7d}: {t*1e3:8.1f} us  {2.0*M*N*K/t/1e9:7.1f} TF/s", flush=True)
elif what == "wide":
    # the same model width (512 = heads x dim_head) at n = 257: 8 x 64 (head-resident attention3.h), then 128-wide head slots (attention.h,
    # two 64-wide halves per head; a narrower head is zero-padded to its slot by the host, so 80 / 96 cost what 128 costs per head)
    for (h, slot, label) in [(8, 64, "8 heads x 64"), (6, 128, "6 heads x 80 (slot 128)"), (5, 128, "5 heads x 96 (slot 128)"), (4, 128, "4 heads x 128")]:
        b, n = 1024, 257
        qkv = torch.randn(b, n, 3 * h * slot, device=dev, dtype=bf)
        mask = torch.ones(b, n, dtype=torch.bool, device=dev)
        sc = (slot if slot == 64 else {6: 80, 5: 96, 4: 128}[h]) ** -0.5
        t = timeit(lambda: ops.attention_fwd(qkv, mask, h, sc, head_dim=slot), iters=10, warm=3)
        out, lse = ops.attention_fwd(qkv, mask, h, sc, head_dim=slot)
        do = torch.randn_like(out)
        tb = timeit(lambda: ops.attention_bwd(qkv, mask, out, do, lse, h, sc, head_dim=slot), iters=10, warm=3)
        print(f"[{tag}] attention b={b} n={n} {label:26s}: fwd {t*1e3:8.1f} us   bwd {tb*1e3:8.1f} us   (per head: fwd {t*1e6/(b*h):6.3f} ns x1e3, bwd {tb*1e6/(b*h):6.3f})", flush=True)
else:
    Mt, Mv = 1024 * 257, 1024 * 33
    for (name, M, N, K, bk, res) in [("ff1 dgrad text", Mt, 512, 4096, True, False), ("ff2 fwd+skip text", Mt, 512, 2048, False, True),
                                     ("qkv dgrad text", Mt, 512, 1536, True, False), ("out fwd text", Mt, 512, 512, False, False),
                                     ("ff1 dgrad vision", Mv, 512, 4096, True, False), ("ff2 fwd+skip vision", Mv, 512, 2048, False, True),
                                     ("qkv dgrad vision", Mv, 512, 1536, True, False), ("qkv fwd vision", Mv, 1536, 512, False, False),
                                     ("ff1 fwd vision", Mv, 4096, 512, False, False)]:
        a = torch.randn(M, K, device=dev, dtype=bf)
        b = torch.randn((K, N) if bk else (N, K), device=dev, dtype=bf)
        r = torch.randn(M, N, device=dev, dtype=bf) if res else None
        out = torch.empty(M, N, device=dev, dtype=bf)
        t = timeit(lambda: ops.gemm(a, b, M, N, K, b_kmajor=bk, residual=r, out=out))
        tiles = ((M + 255) // 256) * (N // 256)
        print(f"[{tag}] {name:22s} M={M:7d} N={N:5d} K={K:5d}: {t*1e3:8.1f} us  {2.0*M*N*K/t/1e9:7.1f} TF/s   ({tiles} tiles = {tiles/256:.2f} rounds)", flush=True)
