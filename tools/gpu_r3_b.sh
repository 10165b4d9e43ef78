#!/bin/bash
# Round-3 second GPU call: the whole GPU suite without -x (wide heads, W = 4 / 8 ranks, live rows ...), the contrastive-head probe at the
# configs[2] per-rank block on the new ring-loop kernels and, through the measurement build, on the round-1 loop.   tools/gpu_r3_b.sh <tag>
TAG=${1:-r03_b}
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1800 python -m pytest tests -m gpu -q --durations=15 ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1
tail -30 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-300
cp gpurun_out/parity_report_gpu.txt gpurun_out/${TAG}_parity_report_gpu.txt 2>/dev/null
(echo "== simloss5.h (g5_run ring loop, whole-line G epilogue): production library"; timeout 300 python tools/probe_sim.py;
 echo "== simloss3.h (round-1 two-stage loop): measurement build, XCLIP_SIM=3"; XCLIP_SIM=3 timeout 300 python tools/probe_sim.py --measure) > gpurun_out/${TAG}_sim_kernels_32k.log 2>&1
cat gpurun_out/${TAG}_sim_kernels_32k.log | cut -c1-200
timeout 600 python bench.py --dcl --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_dcl4096.log 2>&1; tail -1 gpurun_out/${TAG}_bench_dcl4096.log | cut -c1-900
