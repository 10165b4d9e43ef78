#!/bin/bash
# Round-3: rows per wave of the narrow-row LayerNorm forward (rows.h ln_fwd_rows_kernel), A/B through the measurement build, then the
# affected tests and the headline line with the product default.
TAG=${1:-r03_k}
mkdir -p gpurun_out
(for R in 0 1 2 4; do echo "== XCLIP_LN_FWD=$R (0 = one row per wave, loads of gain / residual behind the reductions)"; XCLIP_LN_FWD=$R timeout 200 python tools/probe_ln.py --measure 2>&1 | grep "dim=512\|dim=1024"; done) > gpurun_out/${TAG}_ln_fwd_rows_per_wave.log 2>&1
cat gpurun_out/${TAG}_ln_fwd_rows_per_wave.log | cut -c1-200
( time timeout 900 python -m pytest tests -m gpu -q -k "layernorm or fixture or default_arch" ) > gpurun_out/${TAG}_pytest_gpu_subset.log 2>&1
tail -4 gpurun_out/${TAG}_pytest_gpu_subset.log | cut -c1-200
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-1300
