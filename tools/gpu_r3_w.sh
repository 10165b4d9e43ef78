#!/bin/bash
# Round-3: filip5's token-mask bytes read unconditionally (four serialized round trips per tile before) -- the fused-FILIP kernel
# tests and the kernel trace of the configs[3] line.
TAG=${1:-r03_w}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_clip_gpu.py -m gpu -q -k "filip" 2>&1 | tail -2 | cut -c1-200 | tee gpurun_out/${TAG}_pytest_gpu_subset.log
cd /tmp
rm -rf /tmp/kt_filip
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt_filip -o kt -- python $R/bench.py --filip --batch 512 --steps 4 --warmup 1 --no-overlap --no-probe --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_filip_traced.log 2>&1
DB=$(find /tmp/kt_filip -name "*.db" | head -1)
(echo "# rocprofv3 --kernel-trace -- python bench.py --filip --batch 512 --steps 4 --warmup 1 --no-overlap --no-probe --no-cpu-baseline   (7 single-stream steps incl. the pre-warm ones; summarised by tools/rocpd_stats.py)"; python $R/tools/rocpd_stats.py $DB 40) > $R/gpurun_out/${TAG}_kernel_stats_filip.txt 2>&1
head -14 $R/gpurun_out/${TAG}_kernel_stats_filip.txt | cut -c1-150
tail -1 $R/gpurun_out/${TAG}_bench_filip_traced.log | cut -c1-300
