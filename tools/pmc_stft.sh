#!/bin/bash
# rocprofv3 PMC passes (separate runs, --kernel-trace only alongside) over tools/run_stft_only.py; summaries via tools/pmc_summary.py
#   usage: tools/pmc_stft.sh <outdir under gpurun_out> <n_fft> <clips> <T> [env assignments...]
#   RUNNER=tools/r04/run_nfk_only.py (an env assignment) profiles psnd_stft_mag_nfk instead
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; NFFT=$2; NCLIP=$3; TT=$4; shift 4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
i=0
for set in "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE TCC_EA0_RDREQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$i -o p --output-format csv -- python $ROOT/${RUNNER:-tools/run_stft_only.py} $NFFT $NCLIP $TT 3 > $OUT/pmc_$i.log 2>&1
done
python $ROOT/tools/pmc_summary.py $OUT ${MATCH:-stft_fwd} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
