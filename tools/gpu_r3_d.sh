#!/bin/bash
# Round-3 fourth GPU call: tests touched since call C (fused FILIP at scale with the workspace canary, live rows, 4 ranks, the full-size
# GEMM gate over every interior epilogue), the configs[3] line again (batched LDS reads in the column scan) with a kernel trace, and the
# text tower in two micro-batch slices now that the slices fork before slice 0 is issued.
TAG=${1:-r03_d}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
( time timeout 1200 python -m pytest tests -m gpu -q -k "filip or live_rows or four_ranks or full_size_every_element" ) > gpurun_out/${TAG}_pytest_gpu_subset.log 2>&1
tail -8 gpurun_out/${TAG}_pytest_gpu_subset.log | cut -c1-300
timeout 300 python bench.py --filip --batch 512 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_filip_fused.log 2>&1; tail -1 gpurun_out/${TAG}_bench_filip_fused.log | cut -c1-400
cd /tmp
rm -rf /tmp/kt_filip
timeout 900 rocprofv3 --kernel-trace -d /tmp/kt_filip -o kt -- python $R/bench.py --filip --batch 512 --steps 4 --warmup 1 --no-overlap --no-probe --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_filip_traced.log 2>&1
DB=$(find /tmp/kt_filip -name "*.db" | head -1)
(echo "# rocprofv3 --kernel-trace -- python bench.py --filip --batch 512 --steps 4 --warmup 1 --no-overlap --no-probe --no-cpu-baseline   (7 single-stream steps incl. the pre-warm ones; fused FILIP forward; summarised by tools/rocpd_stats.py)"; python $R/tools/rocpd_stats.py $DB 40) > $R/gpurun_out/${TAG}_kernel_stats_filip.txt 2>&1
grep -n "filip" $R/gpurun_out/${TAG}_kernel_stats_filip.txt | cut -c1-170
cd $R
for S in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --text-slices $S > gpurun_out/${TAG}_bench_text_slices_$S.log 2>&1; tail -1 gpurun_out/${TAG}_bench_text_slices_$S.log | cut -c1-330
done
