#!/bin/bash
# Round-3: the contrastive-head G kernel in two launches (simloss5.h) -- tests, the configs[2] block probe (product, then XCLIP_SIM=3 in
# the measurement build for the same-box A/B), the configs[2] per-GPU line.
TAG=${1:-r03_o}
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -q -k "simloss or nt_xent or simclr or default_arch or fixture or pluggable or dist" ) > gpurun_out/${TAG}_pytest_gpu_subset.log 2>&1
tail -5 gpurun_out/${TAG}_pytest_gpu_subset.log | cut -c1-250
(echo "== product: sim5_lse forward; G = sim5_grad_fast (interior tiles, ring loop) + sim5_grad_edge (tile list)"; timeout 300 python tools/probe_sim.py;
 echo "== measurement build, XCLIP_SIM=3: simloss3.h for both (round-1 loop, G in one launch)"; XCLIP_SIM=3 timeout 300 python tools/probe_sim.py --measure) > gpurun_out/${TAG}_sim_kernels_32k.log 2>&1
grep -v amdgpu gpurun_out/${TAG}_sim_kernels_32k.log | cut -c1-200
timeout 600 python bench.py --dcl --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_dcl4096.log 2>&1; tail -1 gpurun_out/${TAG}_bench_dcl4096.log | cut -c1-400
