#!/bin/bash
# Round-3: HBM-side traffic of the contrastive-head kernels at the configs[2] block (4096 x 32768 x 512): FETCH_SIZE / WRITE_SIZE in
# separate rocprofv3 passes of tools/probe_sim.py --g-only (algorithmic: forward reads 37.7 MB; G reads 37.7 MB + writes 268.4 MB).
TAG=${1:-r03_x}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/tools/probe_sim.py --g-only > /tmp/pmc_$c.log 2>&1
done
F=$(find /tmp/pmc_FETCH_SIZE -name "p_counter_collection.csv" | head -1); W=$(find /tmp/pmc_WRITE_SIZE -name "p_counter_collection.csv" | head -1)
(echo "# HBM-side traffic per kernel, tools/probe_sim.py --g-only (4096 x 32768 x 512, bf16): rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes";
 echo "# algorithmic: sim5_lse_kernel reads 37.7 MB; sim5_grad_fast_kernel reads 37.7 MB + writes 268.4 MB"; python $R/tools/pmc_traffic.py $F $W) > $R/gpurun_out/${TAG}_sim_hbm_traffic_pmc.txt 2>&1
head -12 $R/gpurun_out/${TAG}_sim_hbm_traffic_pmc.txt | cut -c1-170
