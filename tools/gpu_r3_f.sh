#!/bin/bash
# Round-3 short re-check: the full-size live-rows test with its loosened self-consistency bar, the dropout tests, and the contrastive-head
# probe after the kernarg re-read of the forward's epilogue parameters.
TAG=${1:-r03_f}
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -q -k "live_rows or dropout or simloss" ) > gpurun_out/${TAG}_pytest_gpu_subset.log 2>&1
tail -6 gpurun_out/${TAG}_pytest_gpu_subset.log | cut -c1-250
timeout 300 python tools/probe_sim.py > gpurun_out/${TAG}_sim_kernels_32k.log 2>&1; head -8 gpurun_out/${TAG}_sim_kernels_32k.log | cut -c1-200
