#!/bin/bash
# Round-3 closing call at HEAD: the configs[2] block probe of the contrastive-head kernels first (20 s), then the whole GPU suite, smoke
# and the headline measurement set (tools/gpu_r3_e.sh).
TAG=${1:-r03_final3}
mkdir -p gpurun_out
timeout 300 python tools/probe_sim.py > gpurun_out/${TAG}_sim_kernels_32k.log 2>&1
grep -v amdgpu gpurun_out/${TAG}_sim_kernels_32k.log | head -6 | cut -c1-200
bash tools/gpu_r3_e.sh ${TAG}
