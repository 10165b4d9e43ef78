"""round 6: the attention backward's generations side by side on one box, measurement build (XCLIP_ATTN_BWD is read once per process: this
script re-runs itself per generation; XCLIP_AB_GENS=5,7,5 picks the order): 7 = attention7.h (attention5.h persistent, the next head's images
requested by asm-issued DMA under this head's stores), 6 = attention6.h (streaming persistent), 5 = attention5.h (single pass, head resident), 3 =
attention3.h (two phases).  Text-layer shape of configs[1] (b = 1024, n = 257, 8 heads, masked) and n = 256; results must agree with each
other to bf16 rounding (checked against generation 5 through a file)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(gen):
    import torch
    from x_clip_amd import _lib, ops
    _lib.use_measurement_build()
    dev = torch.device("cuda:0")

    def timeit(fn, iters=20, warm=5):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters

    for (b, n, h) in [(1024, 257, 8), (1024, 256, 8), (256, 257, 8), (4096, 257, 8)]:
        g = torch.Generator(device="cpu").manual_seed(5)
        qkv = torch.randn(b, n, 3 * h * 64, generator=g).to(torch.bfloat16).to(dev)
        mask = torch.ones(b, n, dtype=torch.bool)
        mask[::7, 200:] = False                                  # some padded keys (the masked path) incl. the tail key
        mask[3::11, 17] = False
        mask = mask.to(dev)
        out, lse = ops.attention_fwd(qkv, mask, h, 0.125)
        do = torch.randn(out.shape, generator=g).to(torch.bfloat16).to(dev)
        d = ops.attention_bwd(qkv, mask, out, do, lse, h, 0.125)
        torch.cuda.synchronize()
        path = f"/tmp/attn_bwd_ref_{b}_{n}.pt"
        if gen == "5":
            torch.save(d.cpu(), path)
            agree = ""
        elif os.path.exists(path):
            r = torch.load(path).to(dev).float()
            err = float((d.float() - r).abs().max()) / float(r.abs().max())
            agree = f"   max |d - d(gen 5)| / scale = {err:.2e}, equal = {bool(torch.equal(d.float(), r))}, finite = {bool(torch.isfinite(d.float()).all())}"
        else:
            agree = ""
        tb = timeit(lambda: ops.attention_bwd(qkv, mask, out, do, lse, h, 0.125))
        # algorithmic bytes: q, k, v, dO, O in, dq, dk, dv out
        gb = (qkv.numel() * 2 * 2 + out.numel() * 2 * 2) / 1e9
        flops = 5 * 2.0 * n * n * 64 * b * h
        print(f"gen {gen}  b={b:5d} n={n:4d} h={h}: bwd {tb*1e3:8.1f} us   {gb/tb*1e3:6.0f} GB/s ({gb/tb*1e3/8000*100:4.1f} % of 8 TB/s)   {flops/tb/1e9:6.1f} TF/s{agree}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for gen in os.environ.get("XCLIP_AB_GENS", "5,7,6,3,7,5").split(","):
            env = dict(os.environ, XCLIP_ATTN_BWD=gen)
            subprocess.run([sys.executable, os.path.abspath(__file__), gen], env=env, check=False)
