"""round 6: patchify (rearrange + PatchDropout gather, x_clip.py:357) at configs[1]'s size -- the 16-byte-piece kernel for three bf16
channels (product) against the element-wise one (measurement build, XCLIP_PATCHIFY_RGB8=0)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from x_clip_amd import _lib, ops
    _lib.use_measurement_build()
    dev = torch.device("cuda:0")
    for (b, size, p, frac) in [(1024, 256, 32, 0.5), (512, 224, 16, 0.5), (1024, 256, 32, 1.0)]:
        img = torch.randn(b, 3, size, size, device=dev).bfloat16()
        npatch = (size // p) ** 2
        nk = int(npatch * frac)
        keep = torch.randn(b, npatch, device=dev).topk(nk, dim=-1).indices.to(torch.int32) if frac < 1 else None
        for _ in range(5):
            out = ops.patchify(img, p, keep)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30):
            ops.patchify(img, p, keep)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / 30
        gb = 2 * out.numel() * 2 / 1e9
        print(f"RGB8={os.environ.get('XCLIP_PATCHIFY_RGB8', '1')}  b={b} image {size} patch {p} keep {frac}: {t * 1e3:7.1f} us   {gb / t * 1e3:6.0f} GB/s (kept patches in + out)   "
              f"checksum {float(out.float().abs().sum()):.6e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for v in ("0", "1", "0", "1"):
            subprocess.run([sys.executable, os.path.abspath(__file__), "x"], env=dict(os.environ, XCLIP_PATCHIFY_RGB8=v), check=False)
