#!/bin/bash
# SQ counters of the kernels whose name contains <substring>, averaged per dispatch, over three rocprofv3 --pmc passes of a command:
#     tools/pmc_kernel.sh <substring> <tag> <command ...>
#     tools/pmc_kernel.sh gemm9 ffn python tools/probe_ffn_fused.py
#     tools/pmc_kernel.sh attn5 attn257 python tools/probe_attn_one.py 257
# (counter passes carry only --kernel-trace: gpurun refuses --pmc together with the other trace domains; tools/pmc_gemm.sh is the one-GEMM-shape form)
SUB=$1; TAG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
  rm -rf /tmp/pmc_$TAG
  ( cd $R && timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_$TAG -o p -- "$@" ) > /dev/null 2>&1 || true
  python - <<EOF
import csv, glob, collections
f = glob.glob('/tmp/pmc_$TAG/**/p_counter_collection.csv', recursive=True)
tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
name = None
for path in f:
    for r in csv.DictReader(open(path)):
        if '$SUB' in r['Kernel_Name']:
            name = r['Kernel_Name']
            tot[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
for k in sorted(tot):
    print(f"$TAG {str(name)[:40]:40s} {k:28s} {tot[k]/cnt[k]:18.0f}  ({cnt[k]} dispatches)")
EOF
done
