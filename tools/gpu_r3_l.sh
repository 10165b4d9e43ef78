#!/bin/bash
# Round-3 last sanity of HEAD on the MI355X (the library was rebuilt for xclip_build_info): fixtures, kernel suite, smoke, headline bench.
TAG=${1:-r03_l}
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_clip_gpu.py -m gpu -q -k "not vit_l_14 and not full_size_step" ) > gpurun_out/${TAG}_pytest_gpu_subset.log 2>&1
tail -4 gpurun_out/${TAG}_pytest_gpu_subset.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); from x_clip_amd import _lib; print(_lib.lib().xclip_build_info().decode())" 2>&1 | tail -2 | tee gpurun_out/${TAG}_smoke.log
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/${TAG}_bench.log 2>&1; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-600
