#!/bin/bash
# SQ counters of one GEMM shape (rocprofv3 --pmc, CSV):   tools/pmc_gemm.sh M N K layout tag
# (counter passes carry only --kernel-trace: gpurun refuses --pmc together with the other trace domains)
set -e
M=$1; N=$2; K=$3; L=$4; TAG=$5
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
  rm -rf /tmp/pmc_$TAG
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_$TAG -o p -- python $GRAFT_REPO_ROOT/tools/probe_gemm_one.py $M $N $K $L 4 > /dev/null 2>&1 || true
  python - <<EOF
import csv, glob, collections
f = glob.glob('/tmp/pmc_$TAG/**/p_counter_collection.csv', recursive=True)
tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
for path in f:
    for r in csv.DictReader(open(path)):
        if 'gemm' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
for k in sorted(tot):
    print(f"$TAG M=$M N=$N K=$K $L  {k:28s} {tot[k]/cnt[k]:16.0f}  ({cnt[k]} dispatches)")
EOF
done
