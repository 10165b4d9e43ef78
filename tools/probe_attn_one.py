"""runs the attention forward + backward a few times (for rocprofv3 --pmc): python tools/probe_attn_one.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_clip_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
b, h = 1024, 8
qkv = torch.randn(b, n, 3 * h * 64, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    out, lse = ops.attention_fwd(qkv, None, h, 0.125)
do = torch.randn_like(out)
for _ in range(3):
    ops.attention_bwd(qkv, None, out, do, lse, h, 0.125)
torch.cuda.synchronize()
