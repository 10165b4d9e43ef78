#!/usr/bin/env python
"""HBM-side traffic per kernel family from two rocprofv3 PMC passes of the same command (CSV output):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d F -o p -- <cmd>
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d W -o p -- <cmd>
    python tools/pmc_traffic.py F/p_counter_collection.csv W/p_counter_collection.csv [gemm_traffic.json [kernel family, default gemm3 [steps of <cmd>]]]
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts the 128-byte requests of wide coalesced reads as 64 B -> x2
(MI355X_MICROARCH.md, HBM section).  Infinity-Cache hits are part of FETCH_SIZE (fabric traffic: an upper bound on HBM reads)."""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("xc::", "").replace("unsigned short", "bf16")[:60]


def load(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                k = short(r["Kernel_Name"])
                tot[k] += float(r["Counter_Value"])
                cnt[k] += 1
    return tot, cnt


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, _ = load(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in fetch:
        n = nf[k]
        rd = 2 * fetch[k] * 1024 / n / 1e6
        wr = write.get(k, 0.0) * 1024 / n / 1e6
        rows.append((k, n, fetch[k] / n, rd, wr, rd + wr))
    rows.sort(key=lambda r: -r[5] * r[1])
    print(f"{'kernel':60s} {'launches':>8s} {'FETCH KiB/launch':>18s} {'reads x2 (MB)':>14s} {'writes (MB)':>12s} {'total (MB)':>11s}")
    for r in rows[:24]:
        print(f"{r[0]:60s} {r[1]:8d} {r[2]:18.0f} {r[3]:14.1f} {r[4]:12.1f} {r[5]:11.1f}")
    fam = sys.argv[4] if len(sys.argv) > 4 else "gemm3"
    # "gemm" = gemm4_kernel + gemm5_kernel + gemm8_kernel ...; gemm9_geglu_bwd_kernel (the product + LayerNorm backward, its own family in bench.py) is not a plain product
    g = [r for r in rows if r[0].startswith(fam) and "_kernel" in r[0] and not r[0].startswith("gemm9")]
    n = sum(r[1] for r in g)
    per = sum(r[5] * r[1] for r in g) / max(n, 1)
    print(f"# {fam} family: {n} launches, {per:.1f} MB per launch (reads x2 + writes)")
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 0          # steps the traced command ran (bench.py: 2 pre-warm + warm-up + timed)
    if steps:
        print(f"# {fam} family: {per * n / steps / 1e3:.2f} GB per step over {steps} steps ({n // steps} kernel launches per step: one xclip_gemm call is 1 - 3 of them)")
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            import os
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from x_clip_amd.ops import GEMM_GENERATION
            json.dump({"kernel_family": fam + "*_kernel", "kernel_generation": GEMM_GENERATION, "workload_tag": os.environ.get("XCLIP_WORKLOAD_TAG", "default-infonce-b1024"),
                       "bytes_per_launch": per * 1e6, "launches": n, "steps": steps, "bytes_per_step": (per * 1e6 * n / steps) if steps else None,
                       "source": "tools/pmc_traffic.py over rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py (FETCH_SIZE x2 gfx950 correction)"}, f, indent=1)


if __name__ == "__main__":
    main()
