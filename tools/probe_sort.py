"""round 6: sort.h's stable radix sort against torch.sort on the step's two sorts (text token ids: 263,168 below 49,408; kept-patch
indices: 32,768 below 64)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from x_clip_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, iters=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for (n, limit) in [(263168, 49408), (32768, 64), (1 << 20, 1 << 18)]:
    ids = torch.randint(0, limit, (n,), device=dev)
    a = ops.sort_ids(ids, limit)
    b = torch.sort(ids, stable=True)
    ok = torch.equal(a[0], b.values) and torch.equal(a[1], b.indices)
    print(f"n = {n:8d} ids < {limit:7d}: sort.h {timeit(lambda: ops.sort_ids(ids, limit)):7.1f} us   torch.sort {timeit(lambda: torch.sort(ids)):7.1f} us   "
          f"torch.sort(stable) {timeit(lambda: torch.sort(ids, stable=True)):7.1f} us   same result: {ok}", flush=True)
