"""K = 512 forward shapes of the text tower under the kernel-selection switch XCLIP_GEMM (3 = gemm3.h, 4 = g4_run, 5 = g5_run; see xclip_api.hip):
    python tools/probe_gemm_k512.py      (one process per setting: the switches are read once per process)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_clip_amd import _lib, ops
_lib.use_measurement_build()       # the XCLIP_* switches below exist only in libxclip_hip_measure.so (python -m x_clip_amd.build --measure)
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("XCLIP_GEMM"))
M = 263168
for (N, K) in [(1536, 512), (4096, 512), (512, 512), (512, 2048)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.gemm(a, b, M, N, K, out=out))
    tiles = (M // 256) * (N // 256)
    print(f"[{tag or 'default'}] NT M={M} N={N} K={K}: {t*1e3:8.1f} us {2*M*N*K/t/1e9:7.1f} TF/s  ({t*1e3*256/tiles:6.2f} us per tile per CU)", flush=True)
