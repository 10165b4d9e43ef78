#!/bin/bash
# SQ counters of the attention kernels at one shape (rocprofv3 --pmc, CSV):   tools/pmc_attn.sh B N H
# (counter passes carry only --kernel-trace: gpurun refuses --pmc together with the other trace domains)
B=$1; N=$2; H=$3
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC" "SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE SQ_WAVES"; do
  rm -rf /tmp/pmc_attn
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_attn -o p -- python $GRAFT_REPO_ROOT/tools/probe_attn_one.py $N > /dev/null 2>&1 || true
  python - <<EOF
import csv, glob, collections
f = glob.glob('/tmp/pmc_attn/**/p_counter_collection.csv', recursive=True)
tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
for path in f:
    for r in csv.DictReader(open(path)):
        if 'attn' in r['Kernel_Name']:
            k = (r['Kernel_Name'].split('(')[0][-28:], r['Counter_Name'])
            tot[k] += float(r['Counter_Value']); cnt[k] += 1
for k in sorted(tot):
    print(f"b=$B n=$N h=$H  {k[0]:28s} {k[1]:28s} {tot[k]/cnt[k]:16.0f}  ({cnt[k]} dispatches)")
EOF
done
