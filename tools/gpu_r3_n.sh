#!/bin/bash
# Round-3: size of the FILIP backward's routing-matrix chunks against the 256 MB Infinity Cache (configs[3], b = 512)
TAG=${1:-r03_n}
mkdir -p gpurun_out
for MB in 1024 512 256 128; do
  XCLIP_FILIP_CHUNK_MB=$MB timeout 300 python bench.py --filip --batch 512 --steps 8 --warmup 2 --no-cpu-baseline --no-probe > gpurun_out/${TAG}_bench_filip_chunk_${MB}mb.log 2>&1
  echo "chunk $MB MB: $(tail -1 gpurun_out/${TAG}_bench_filip_chunk_${MB}mb.log | cut -c1-260)"
done
