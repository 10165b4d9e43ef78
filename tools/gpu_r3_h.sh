#!/bin/bash
# Round-3: the FILIP backward's routing kernel after its rewrite -- the FILIP tests, the configs[3] line and its kernel trace.
TAG=${1:-r03_h}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
( time timeout 900 python -m pytest tests -m gpu -q -k "filip" ) > gpurun_out/${TAG}_pytest_gpu_subset.log 2>&1
tail -5 gpurun_out/${TAG}_pytest_gpu_subset.log | cut -c1-250
timeout 300 python bench.py --filip --batch 512 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_filip_fused.log 2>&1; tail -1 gpurun_out/${TAG}_bench_filip_fused.log | cut -c1-400
cd /tmp
rm -rf /tmp/kt_filip
timeout 900 rocprofv3 --kernel-trace -d /tmp/kt_filip -o kt -- python $R/bench.py --filip --batch 512 --steps 4 --warmup 1 --no-overlap --no-probe --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_filip_traced.log 2>&1
DB=$(find /tmp/kt_filip -name "*.db" | head -1)
(echo "# rocprofv3 --kernel-trace -- python bench.py --filip --batch 512 --steps 4 --warmup 1 --no-overlap --no-probe --no-cpu-baseline   (7 single-stream steps incl. the pre-warm ones; summarised by tools/rocpd_stats.py)"; python $R/tools/rocpd_stats.py $DB 40) > $R/gpurun_out/${TAG}_kernel_stats_filip.txt 2>&1
head -16 $R/gpurun_out/${TAG}_kernel_stats_filip.txt | cut -c1-170
