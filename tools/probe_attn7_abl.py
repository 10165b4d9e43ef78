"""round 6: where a head of attention7.h (attention5.h persistent, next images requested under the stores) goes -- the measurement build's
ablation mask XCLIP_ATTN7_ABL (results are garbage): 1 no stores, 2 the images requested at the top of the head (no overlap with the stores),
4 no delta pass, 8 no pairs"""
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
            d = ops.attention_bwd(qkv, mask, out, do, lse, h, 0.125)
        torch.cuda.synchronize()
        chk = float(d.float().abs().sum())                       # (equal across variants that must not change the result)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            ops.attention_bwd(qkv, mask, out, do, lse, h, 0.125)
        e.record()
        torch.cuda.synchronize()
        print(f"XCLIP_ATTN_BWD={os.environ.get('XCLIP_ATTN_BWD', '7')} XCLIP_ATTN7_ABL={os.environ.get('XCLIP_ATTN7_ABL', '0'):>3s}  b={b} n={n}: {s.elapsed_time(e) / 20 * 1e3:8.1f} us   |d|_1 = {chk:.6e}  VAR5={os.environ.get('XCLIP_ATTN5_VAR', '0')}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "var5":          # attention5.h's variant mask: 1 quarter-line stores (round 5), 2 images before the delta rows (round 5), 4 no stores
        for var in (3, 0, 1, 2, 3, 0, 4):
            subprocess.run([sys.executable, os.path.abspath(__file__), "x"], check=False, env=dict(os.environ, XCLIP_ATTN_BWD="5", XCLIP_ATTN5_VAR=str(var)))
    elif len(sys.argv) > 1:
        child()
    else:
        for abl in (0, 1, 2, 4, 8, 1 + 4, 1 + 8, 4 + 8, 1 + 4 + 8, 2 + 1, 0):
            subprocess.run([sys.executable, os.path.abspath(__file__), "x"], env=dict(os.environ, XCLIP_ATTN_BWD="7", XCLIP_ATTN7_ABL=str(abl)), check=False)
        subprocess.run([sys.executable, os.path.abspath(__file__), "x"], env=dict(os.environ, XCLIP_ATTN_BWD="5"), check=False)
