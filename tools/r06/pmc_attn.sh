#!/bin/bash
# round 6: counter passes over the config-4 leg (bench.py --leg config4), summary of the attention kernels -> gpurun_out/r06/pmc_attn/
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06/pmc_attn; RAW=/tmp/pmc_attn; mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_IFETCH SQ_WAIT_IFETCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $RAW/pmc_$i -o p --output-format csv -- python $ROOT/bench.py --leg config4 > $RAW/pmc_$i.log 2>&1; tail -2 $RAW/pmc_$i.log | cut -c1-300 > $OUT/pmc_$i.tail
done
python - <<'PY' > $OUT/summary.txt
import csv, glob, os, collections, re
root = '/tmp/pmc_attn'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + '/pmc_*/*counter_collection.csv')):
    for row in csv.DictReader(open(f)):
        n = row['Kernel_Name']
        if 'attn_' not in n and 'gemm_bf16' not in n:
            continue
        m = re.search(r'(attn_\w+_kernel<\d+, \w+>|attn_\w+_kernel|gemm_bf16_kernel<[\w, ]+>)', n)
        agg[m.group(1) if m else n[:50]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, c in sorted(agg.items()):
    print('==', k)
    for name, v in sorted(c.items()):
        print('   %-34s %16.0f  (n=%d)' % (name, sum(v) / len(v), len(v)))
PY
cat $OUT/summary.txt | head -150
