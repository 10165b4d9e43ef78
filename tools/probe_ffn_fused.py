"""xclip_ffn_dgrad_geglu (gemm9.h) alone at the towers' sizes: microseconds per call, HBM rate over its algorithmic bytes, and MFMA rate.
    python tools/probe_ffn_fused.py [pmc]               the product library (pmc: a few launches of the text-tower size, for counter passes)
    XCLIP_GEMM9_ABL=<n> python tools/probe_ffn_fused.py measure     libxclip_hip_measure.so with an ablation of the epilogue (gemm9.h: 1 asm line
                                                                    stores, 2 no GELU arithmetic, 4 no line loads behind the first group, 8 no line
                                                                    stores, 14 all three): where a tile's time goes.  Ablated results are garbage."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x_clip_amd import _lib, ops  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "measure":
    _lib.use_measurement_build()
DEV = torch.device("cuda:0")


def run(M, F, D, iters=10, rounds=5):
    g = torch.Generator(device="cpu").manual_seed(0)
    dout = (torch.randn(M, D, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    w2 = (torch.randn(D, F, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    x = torch.randn(M, 2 * F, generator=g).to(torch.bfloat16).to(DEV)
    gamma = torch.randn(F, generator=g).to(torch.bfloat16).to(DEV)
    mean = torch.zeros(M, dtype=torch.float32, device=DEV)
    rstd = torch.ones(M, dtype=torch.float32, device=DEV)
    x2, x1 = torch.randn(M, D, device=DEV).to(torch.bfloat16), torch.randn(M, D, device=DEV).to(torch.bfloat16)
    assert ops.ffn_dgrad_geglu_ok(M, F, D, torch.bfloat16)
    call = lambda: ops.ffn_dgrad_geglu(dout, w2, x, gamma, mean, rstd, x2, x1)   # noqa: E731
    for _ in range(2):
        call()
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            call()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e3)
    us = sorted(ts)[len(ts) // 2]
    gb = M * (4 * F + D) * 2 / 1e9                      # x [M, 2F] read, dx [M, 2F] written, dout [M, D] read
    print(f"ffn_dgrad_geglu M={M:6d} F={F} D={D} ABL={os.environ.get('XCLIP_GEMM9_ABL', '0')}: {us:8.1f} us   {gb / us * 1e6:6.0f} GB/s over {gb:.2f} GB"
          f"   {2.0 * M * F * D / us * 1e-6:6.0f} TF/s   ({M // 256 * (F // 256)} tiles: {us / -(-(M // 256 * (F // 256)) // 256):.1f} us per round of 256)", flush=True)


if __name__ == "__main__":
    if "pmc" in sys.argv:                                   # a few launches of the text-tower size only (counter passes: tools/pmc_kernel.sh)
        run(263168, 2048, 512, iters=2, rounds=1)
    else:
        run(263168, 2048, 512)
        run(32768, 2048, 512)
