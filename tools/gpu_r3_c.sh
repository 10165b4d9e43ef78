#!/bin/bash
# Round-3 third GPU call: the tests touched since call B, the configs[3] (FILIP) line with the fused forward (+ kernel trace, + the
# chunked form through XCLIP_FILIP_FUSED=0 for the A/B), the configs[4] (ViT-L) line with the vision tower in two slices.
TAG=${1:-r03_c}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
( time timeout 1200 python -m pytest tests -m gpu -q -k "filip or live_rows or four_ranks or wide_heads or simloss or micro or rotary" ) > gpurun_out/${TAG}_pytest_gpu_subset.log 2>&1
tail -12 gpurun_out/${TAG}_pytest_gpu_subset.log | cut -c1-300
timeout 300 python bench.py --filip --batch 512 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_filip_fused.log 2>&1; tail -1 gpurun_out/${TAG}_bench_filip_fused.log | cut -c1-700
XCLIP_FILIP_FUSED=0 timeout 300 python bench.py --filip --batch 512 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_filip_chunked.log 2>&1; tail -1 gpurun_out/${TAG}_bench_filip_chunked.log | cut -c1-700
cd /tmp
rm -rf /tmp/kt_filip
timeout 900 rocprofv3 --kernel-trace -d /tmp/kt_filip -o kt -- python $R/bench.py --filip --batch 512 --steps 4 --warmup 1 --no-overlap --no-probe --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_filip_traced.log 2>&1
DB=$(find /tmp/kt_filip -name "*.db" | head -1)
(echo "# rocprofv3 --kernel-trace -- python bench.py --filip --batch 512 --steps 4 --warmup 1 --no-overlap --no-probe --no-cpu-baseline   (7 single-stream steps incl. the pre-warm ones; fused FILIP forward; summarised by tools/rocpd_stats.py)"; python $R/tools/rocpd_stats.py $DB 40) > $R/gpurun_out/${TAG}_kernel_stats_filip.txt 2>&1
head -14 $R/gpurun_out/${TAG}_kernel_stats_filip.txt | cut -c1-170
cd $R
timeout 600 python bench.py --config vitl --batch 2048 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_vitl_2slices.log 2>&1; tail -1 gpurun_out/${TAG}_bench_vitl_2slices.log | cut -c1-1100
