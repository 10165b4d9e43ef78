#!/bin/bash
# Round-3 first GPU call: the whole GPU suite (incl. the real-architecture oracle tests, the full-size live-rows test, W = 4 / 8 ranks on
# cuda:0), smoke, the headline bench, kernel-trace summaries of the configs[4] (vitl) and configs[3] (filip) lines, and an 8-process
# bench dry run on one device with gloo.   tools/gpu_r3_a.sh <tag>
TAG=${1:-r03_a}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
( time timeout 1500 python -m pytest tests -m gpu -q --durations=12 ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1
tail -25 gpurun_out/${TAG}_pytest_gpu.log
cp gpurun_out/parity_report_gpu.txt gpurun_out/${TAG}_parity_report_gpu.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${TAG}_smoke.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench.log 2>&1; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-1500
cd /tmp
for CFG in "vitl:--config vitl --batch 2048 --steps 1 --warmup 0" "filip:--filip --batch 512 --steps 4 --warmup 1"; do
  NAME=${CFG%%:*}; ARGS=${CFG#*:}
  rm -rf /tmp/kt_$NAME
  timeout 900 rocprofv3 --kernel-trace -d /tmp/kt_$NAME -o kt -- python $R/bench.py $ARGS --no-overlap --no-probe --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_${NAME}_traced.log 2>&1
  DB=$(find /tmp/kt_$NAME -name "*.db" | head -1)
  (echo "# rocprofv3 --kernel-trace -- python bench.py $ARGS --no-overlap --no-probe --no-cpu-baseline   (all steps of the run incl. the 2 pre-warm steps, single stream; summarised by tools/rocpd_stats.py)"; python $R/tools/rocpd_stats.py $DB 40) > $R/gpurun_out/${TAG}_kernel_stats_${NAME}.txt 2>&1
  head -16 $R/gpurun_out/${TAG}_kernel_stats_${NAME}.txt | cut -c1-170
  tail -1 $R/gpurun_out/${TAG}_bench_${NAME}_traced.log | cut -c1-600
done
cd $R
# untraced lines of the same two configurations
timeout 300 python bench.py --filip --batch 512 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_filip.log 2>&1; tail -1 gpurun_out/${TAG}_bench_filip.log | cut -c1-900
# the N = 8 launch line of the driver, on ONE device with gloo: the multi-rank bench path end to end (GradSync, gathers, barrier, max-over-ranks)
XCLIP_BENCH_ONE_DEVICE=1 XCLIP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 \
  bench.py --gpus 8 --batch 256 --steps 3 --warmup 1 --no-probe > gpurun_out/${TAG}_bench_8proc_one_device_gloo.log 2>&1
tail -1 gpurun_out/${TAG}_bench_8proc_one_device_gloo.log | cut -c1-900
