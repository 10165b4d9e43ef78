"""round 6: where a step of attention6.h goes -- the measurement build's ablation mask XCLIP_ATTN6_ABL (results are garbage): 1 no atomics,
2 constant fixed-point scale, 4 no delta, 8 no dQ partial at all, 16 no finishing pass, 32 no tail-key work, 64 no requests after the cold start"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from x_clip_amd import _lib, ops
    _lib.use_measurement_build()
    dev = torch.device("cuda:0")
    for (b, n, h) in [(1024, 257, 8), (1024, 256, 8)]:
        g = torch.Generator(device="cpu").manual_seed(5)
        qkv = torch.randn(b, n, 3 * h * 64, generator=g).to(torch.bfloat16).to(dev)
        mask = torch.ones(b, n, dtype=torch.bool, device=dev)
        out, lse = ops.attention_fwd(qkv, mask, h, 0.125)
        do = torch.randn(out.shape, generator=g).to(torch.bfloat16).to(dev)
        for _ in range(5):
            ops.attention_bwd(qkv, mask, out, do, lse, h, 0.125)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            ops.attention_bwd(qkv, mask, out, do, lse, h, 0.125)
        e.record()
        torch.cuda.synchronize()
        print(f"XCLIP_ATTN_BWD={os.environ.get('XCLIP_ATTN_BWD', '6')} XCLIP_ATTN6_ABL={os.environ.get('XCLIP_ATTN6_ABL', '0'):>3s}  b={b} n={n}: {s.elapsed_time(e) / 20 * 1e3:8.1f} us", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for abl in (0, 1, 2, 4, 8, 16, 32, 64, 1 + 2, 8 + 16, 8 + 16 + 32 + 4 + 2, 127):
            subprocess.run([sys.executable, os.path.abspath(__file__), "x"], env=dict(os.environ, XCLIP_ATTN_BWD="6", XCLIP_ATTN6_ABL=str(abl)), check=False)
        subprocess.run([sys.executable, os.path.abspath(__file__), "x"], env=dict(os.environ, XCLIP_ATTN_BWD="5"), check=False)
