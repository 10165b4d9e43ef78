#!/bin/bash
# Round-3 closing GPU call: the whole GPU suite, smoke, then the round's measurement set of the headline line (tools/gpu_round.sh:
# bench with the CPU baseline, kernel-trace summary of single-stream steps, FETCH_SIZE / WRITE_SIZE passes -> roofline.traffic).
TAG=${1:-r03_final}
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q --durations=10 ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1
tail -22 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-250
cp gpurun_out/parity_report_gpu.txt gpurun_out/${TAG}_parity_report_gpu.txt 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/${TAG}_smoke.log
bash tools/gpu_round.sh ${TAG} skip-extra
