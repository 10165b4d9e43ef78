#!/bin/bash
# Round-3: the contrastive-head G kernel at the configs[2] block, product and (measurement build, VARIANTS) 7 = its full-tile launch
# alone, 8 = its edge launch alone, 3 = the round-1 kernels on the same box (6 was the pack_lines / store_lines form of run q: no
# faster, spills since); then the simloss GPU tests.
TAG=${1:-r03_q}
mkdir -p gpurun_out
(echo "== product"; timeout 300 python tools/probe_sim.py --g-only
 for v in ${VARIANTS:-7 8 3}; do echo "== measurement build, XCLIP_SIM=$v"; XCLIP_SIM=$v timeout 300 python tools/probe_sim.py --measure --g-only; done) > gpurun_out/${TAG}_sim_g_variants.log 2>&1
grep -v amdgpu gpurun_out/${TAG}_sim_g_variants.log | cut -c1-220
timeout 600 python -m pytest tests -m gpu -q -k "simloss" 2>&1 | tail -2
