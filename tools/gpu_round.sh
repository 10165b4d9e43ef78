#!/bin/bash
# The measurement set of a round, on the GPU box (gpurun):  tools/gpu_round.sh <tag> [skip-extra]
#   bench default (with CPU baseline) / vitl / dcl 4096; rocprofv3 kernel-trace summary of the default bench on one stream;
#   FETCH_SIZE / WRITE_SIZE passes -> HBM-side traffic of the GEMM family.  Everything lands in gpurun_out/<tag>_*.
TAG=$1
mkdir -p gpurun_out
python bench.py --steps 10 --warmup 3 > gpurun_out/${TAG}_bench.log 2>&1; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-1200
if [ -z "$2" ]; then
  python bench.py --config vitl --batch 2048 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_vitl.log 2>&1; tail -1 gpurun_out/${TAG}_bench_vitl.log | cut -c1-900
  python bench.py --dcl --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_dcl4096.log 2>&1; tail -1 gpurun_out/${TAG}_bench_dcl4096.log | cut -c1-900
fi
export TMPDIR=/tmp
R=$PWD
cd /tmp
rm -rf /tmp/kt; rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --steps 5 --warmup 1 --no-overlap --no-probe --no-cpu-baseline > /dev/null 2>&1
DB=$(find /tmp/kt -name "*.db" | head -1)
(echo "# rocprofv3 --kernel-trace -- python bench.py --steps 5 --warmup 1 --no-overlap --no-probe --no-cpu-baseline   (8 single-stream steps incl. the 2 pre-warm steps; summarised by tools/rocpd_stats.py)"; python $R/tools/rocpd_stats.py $DB 45) > $R/gpurun_out/${TAG}_kernel_stats.txt 2>&1
head -14 $R/gpurun_out/${TAG}_kernel_stats.txt | cut -c1-190
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c; rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-probe --no-overlap > /dev/null 2>&1
done
F=$(find /tmp/pmc_FETCH_SIZE -name "p_counter_collection.csv" | head -1); W=$(find /tmp/pmc_WRITE_SIZE -name "p_counter_collection.csv" | head -1)
(echo "# HBM-side traffic per kernel, bench.py --steps 1 --warmup 1 --no-overlap (b=1024): rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes"; python $R/tools/pmc_traffic.py $F $W $R/gpurun_out/${TAG}_gemm_traffic.json gemm) > $R/gpurun_out/${TAG}_hbm_traffic_pmc.txt 2>&1
tail -3 $R/gpurun_out/${TAG}_hbm_traffic_pmc.txt
