"""round 6: the head-resident forward with the first query rows requested before the K / V images (product) against the round-5 order
(measurement build, XCLIP_ATTN_ABL=8: behind the barrier) -- one box, alternated; the outputs must be the same bits."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from x_clip_amd import _lib, ops
    _lib.use_measurement_build()
    dev = torch.device("cuda:0")
    for (b, n, h, masked) in [(1024, 257, 8, True), (1024, 256, 8, True), (1024, 33, 8, False), (1024, 65, 8, False)]:
        g = torch.Generator(device="cpu").manual_seed(5)
        qkv = torch.randn(b, n, 3 * h * 64, generator=g).to(torch.bfloat16).to(dev)
        mask = None
        if masked:
            mask = torch.ones(b, n, dtype=torch.bool)
            mask[::7, 200:] = False
            mask = mask.to(dev)
        for _ in range(10):
            out, lse = ops.attention_fwd(qkv, mask, h, 0.125)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(40):
            ops.attention_fwd(qkv, mask, h, 0.125)
        e.record()
        torch.cuda.synchronize()
        chk = float(out.float().abs().sum()) + float(lse.abs().sum())
        print(f"XCLIP_ATTN_ABL={os.environ.get('XCLIP_ATTN_ABL', '0')}  b={b} n={n}: fwd {s.elapsed_time(e) / 40 * 1e3:8.1f} us   checksum {chk:.8e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for abl in (8, 0, 8, 0):
            subprocess.run([sys.executable, os.path.abspath(__file__), "x"], env=dict(os.environ, XCLIP_ATTN_ABL=str(abl)), check=False)
