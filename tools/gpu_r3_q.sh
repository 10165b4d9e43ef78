#!/bin/bash
# Round-3: the interior G launch with its stores left in flight over the next tile's first K step; measurement build: 6 = through the
# pack_lines / store_lines pair (early A pieces), 7 = interior launch alone, 8 = edge launch alone.
TAG=${1:-r03_q}
mkdir -p gpurun_out
(echo "== product"; timeout 300 python tools/probe_sim.py --g-only
 for v in ${VARIANTS:-6 7 8 3}; do echo "== measurement build, XCLIP_SIM=$v"; XCLIP_SIM=$v timeout 300 python tools/probe_sim.py --measure --g-only; done) > gpurun_out/${TAG}_sim_g_variants.log 2>&1
grep -v amdgpu gpurun_out/${TAG}_sim_g_variants.log | cut -c1-220
timeout 600 python -m pytest tests -m gpu -q -k "simloss" 2>&1 | tail -2
