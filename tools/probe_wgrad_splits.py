"""round 6: the vision tower's weight gradients (K = 33,792 rows) against the number of K slices -- measurement build, XCLIP_GEMM_SPLITS=<n>
(0 = the policy xclip_api.hip gemm2_splits).  Each slice writes an fp32 slab of the whole output, so few-tile outputs pay 64 slabs."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    from x_clip_amd import _lib, ops
    _lib.use_measurement_build()
    dev = torch.device("cuda:0")
    for (M, N, K) in [(512, 512, 33792), (1536, 512, 33792), (4096, 512, 33792), (512, 2048, 33792), (512, 512, 263168), (1536, 512, 263168)]:
        a = torch.randn(K, M, device=dev, dtype=torch.bfloat16)
        b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for _ in range(5):
            ops.gemm(a, b, M, N, K, True, True, out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30):
            ops.gemm(a, b, M, N, K, True, True, out=out)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / 30
        ws = _lib.lib().xclip_gemm_workspace_bytes(M, N, K, 1)
        print(f"SPLITS={os.environ.get('XCLIP_GEMM_SPLITS', '0'):>3s}  wgrad {M:5d} x {N:5d} x {K:6d}: {t * 1e3:8.1f} us  {2.0 * M * N * K / t / 1e9:7.1f} TF/s   slices {ws // (M * N * 4):3d}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        for n in (0, 4, 8, 16, 32, 64, 0):
            subprocess.run([sys.executable, os.path.abspath(__file__), "x"], env=dict(os.environ, XCLIP_GEMM_SPLITS=str(n)), check=False)
