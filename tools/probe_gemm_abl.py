"""per-K-step cost of the NT GEMM with parts of the kernel switched off: XCLIP_GEMM5_ABL=<mask> python tools/probe_gemm_abl.py
(masks: gemm4.h g5_run)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_clip_amd import _lib, ops
_lib.use_measurement_build()       # the XCLIP_* switches below exist only in libxclip_hip_measure.so (python -m x_clip_amd.build --measure)
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for (M, N, K) in [(263168, 1536, 512), (263168, 512, 2048), (263168, 4096, 512), (263168, 512, 512)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm(a, b, M, N, K, out=out))
    if int(os.environ.get('XCLIP_GEMM5_ABL', '0')) & 512:
        torch.cuda.synchronize()
        c = out.view(-1)[:8].view(torch.int64).tolist()
        print(f"    work-group 0: {c[0]} shader cycles in {c[1] * 10} ns -> {c[0] / (c[1] * 10):.3f} GHz")
    if int(os.environ.get('XCLIP_GEMM5_ABL', '0')) & 2048:
        torch.cuda.synchronize()
        st = out.view(-1)[:4 * (16 + 96)].view(torch.int64)[16:16 + 96].view(8, 12).tolist()
        for w, r in enumerate(st):
            steps = [r[q + 1] - r[q] for q in range(8)]
            print(f"    wave {w}: K steps of tile 2 {steps}  last step end -> next tile start (epilogue) {r[10] - r[8]}  next tile's first step {r[11] - r[10]}")
    print(f"ABL={os.environ.get('XCLIP_GEMM5_ABL','0'):>2s}  M={M} N={N} K={K}: {t*1e3:8.1f} us  ({t*1e3*256/((M//256)*(N//256)*(K//64)):6.3f} us per K step per CU)", flush=True)
