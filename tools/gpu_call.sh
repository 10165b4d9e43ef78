#!/bin/bash
# One GPU-box call (gpurun) = a tag and a list of tasks; everything lands in gpurun_out/<tag>_*.   tools/gpu_call.sh <tag> <task> ...
#   tests[=<pytest -k expression>]   pytest tests -m gpu (whole GPU suite, or a selection) + the parity report
#   smoke                            __graft_entry__.smoke()
#   bench[=<name>:<bench.py args>]   one bench.py line -> <tag>_bench[_<name>].log        (default: the driver's line, --steps 20 --warmup 5)
#   trace[=<name>:<bench.py args>]   rocprofv3 --kernel-trace of bench.py <args> on one stream -> <tag>_kernel_stats[_<name>].txt
#   traffic                          FETCH_SIZE / WRITE_SIZE passes of the default bench -> <tag>_hbm_traffic_pmc.txt, <tag>_gemm_traffic.json
#   py=<name>:<script> <args>        python tools/<script> <args> -> <tag>_<name>.log   (probes; set XCLIP_* variables in front of the call)
#   sq=<name>:<M> <N> <K> <layout>   SQ counter passes of one GEMM shape (tools/pmc_gemm.sh) -> <tag>_sq_<name>.txt
# Replaces round 3's sixteen one-shot tools/gpu_r3_*.sh (each a particular list of these tasks).  Examples:
#   tools/gpu_call.sh r04_a tests smoke bench trace traffic
#   tools/gpu_call.sh r04_b "tests=simloss or gemm" "bench=dcl4096:--dcl --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline" "py=sim:probe_sim.py --g-only"
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
for TASK in "$@"; do
  KIND=${TASK%%=*}; ARG=""; [ "$KIND" != "$TASK" ] && ARG=${TASK#*=}
  NAME=""; REST="$ARG"
  case "$ARG" in *:*) NAME=${ARG%%:*}; REST=${ARG#*:};; esac
  SUF=""; [ -n "$NAME" ] && SUF="_$NAME"
  cd $R
  case $KIND in
    tests)
      if [ -n "$ARG" ]; then ( time timeout 1500 python -m pytest tests -m gpu -q -k "$ARG" --durations=8 ) > gpurun_out/${TAG}_pytest_gpu_subset.log 2>&1; tail -12 gpurun_out/${TAG}_pytest_gpu_subset.log
      else ( time timeout ${TESTS_TIMEOUT:-1700} python -m pytest tests -m gpu -q --durations=12 ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -22 gpurun_out/${TAG}_pytest_gpu.log; fi
      cp gpurun_out/parity_report_gpu.txt gpurun_out/${TAG}_parity_report_gpu.txt 2>/dev/null;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${TAG}_smoke.log;;
    bench)
      [ -z "$ARG" ] && REST="--steps 20 --warmup 5"
      timeout 900 python bench.py $REST > gpurun_out/${TAG}_bench${SUF}.log 2>&1; tail -1 gpurun_out/${TAG}_bench${SUF}.log | cut -c1-2600;;
    trace)
      [ -z "$ARG" ] && REST="--steps 5 --warmup 1"
      cd /tmp; rm -rf /tmp/kt$SUF
      timeout 1200 rocprofv3 --kernel-trace -d /tmp/kt$SUF -o kt -- python $R/bench.py $REST --no-overlap --no-probe --no-cpu-baseline --no-dense-compare > $R/gpurun_out/${TAG}_bench${SUF}_traced.log 2>&1
      DB=$(find /tmp/kt$SUF -name "*.db" | head -1)
      (echo "# rocprofv3 --kernel-trace -- python bench.py $REST --no-overlap --no-probe --no-cpu-baseline --no-dense-compare   (every step of the run incl. the 2 pre-warm steps, single stream; summarised by tools/rocpd_stats.py)"; python $R/tools/rocpd_stats.py $DB 45) > $R/gpurun_out/${TAG}_kernel_stats${SUF}.txt 2>&1
      head -16 $R/gpurun_out/${TAG}_kernel_stats${SUF}.txt | cut -c1-190;;
    traffic)
      cd /tmp
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_$c; timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-probe --no-overlap --no-dense-compare > /dev/null 2>&1
      done
      F=$(find /tmp/pmc_FETCH_SIZE -name "p_counter_collection.csv" | head -1); W=$(find /tmp/pmc_WRITE_SIZE -name "p_counter_collection.csv" | head -1)
      (echo "# HBM-side traffic per kernel, bench.py --steps 1 --warmup 1 --no-overlap --no-dense-compare (b=1024): rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes"; python $R/tools/pmc_traffic.py $F $W $R/gpurun_out/${TAG}_gemm_traffic.json gemm 4) > $R/gpurun_out/${TAG}_hbm_traffic_pmc.txt 2>&1
      tail -3 $R/gpurun_out/${TAG}_hbm_traffic_pmc.txt;;
    py)
      ( timeout 900 python tools/$REST ) > gpurun_out/${TAG}${SUF}.log 2>&1; grep -v amdgpu gpurun_out/${TAG}${SUF}.log | tail -40 | cut -c1-230;;
    sq)
      ( GRAFT_REPO_ROOT=$R timeout 900 bash tools/pmc_gemm.sh $REST $NAME ) > gpurun_out/${TAG}_sq${SUF}.txt 2>&1; tail -24 gpurun_out/${TAG}_sq${SUF}.txt | cut -c1-160;;
    *) echo "unknown task $TASK";;
  esac
done
