#!/bin/bash
# Round-3 re-check after the last kernel-side changes (fused FILIP for segments >= 32 tokens, one GEMM loop per layout in the product
# library, reference fixtures for wide heads): the affected tests, the configs[3] line, the headline line.
TAG=${1:-r03_g}
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -k "filip or gemm or fixture or wide or live_rows" ) > gpurun_out/${TAG}_pytest_gpu_subset.log 2>&1
tail -6 gpurun_out/${TAG}_pytest_gpu_subset.log | cut -c1-250
timeout 300 python bench.py --filip --batch 512 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_filip_fused.log 2>&1; tail -1 gpurun_out/${TAG}_bench_filip_fused.log | cut -c1-400
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-1300
