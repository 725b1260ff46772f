#!/bin/bash
# rocprofv3 PMC passes (separate runs, --kernel-trace only alongside) over any command; summaries via tools/pmc_summary.py
#   usage: tools/pmc_cmd.sh <outdir under gpurun_out> <kernel substrings, comma separated> <command ...>
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; MATCH=$2; shift 2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32" \
           "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$i -o p --output-format csv -- "$@" > $OUT/pmc_$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/pmc_$i.log)"
done
for m in ${MATCH//,/ }; do
  echo "== $m"; python $ROOT/tools/pmc_summary.py $OUT $m
done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
