import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_clip_amd import _lib, ops
_lib.use_measurement_build()       # the XCLIP_* switches below exist only in libxclip_hip_measure.so (python -m x_clip_amd.build --measure)
dev = torch.device("cuda")
def timeit(fn, n=20):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (b, n, h) in [(1024, 257, 8), (1024, 256, 8)]:
    qkv = torch.randn(b, n, 3 * h * 64, device=dev, dtype=torch.bfloat16)
    out, lse = ops.attention_fwd(qkv, None, h, 0.125)
    do = torch.randn_like(out)
    t = timeit(lambda: ops.attention_bwd(qkv, None, out, do, lse, h, 0.125))
    print(f"ATTN_ABL={os.environ.get('XCLIP_ATTN_ABL','0')} b={b} n={n}: bwd {t:8.1f} us", flush=True)
