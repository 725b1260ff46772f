#!/bin/bash
# one rocprofv3 PMC pass (write-side counters) + kernel trace over tools/run_stft_only.py
#   usage: tools/pmc_write.sh <outdir under gpurun_out> <n_fft> <clips> <T> [env assignments...]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; NFFT=$2; NCLIP=$3; TT=$4; shift 4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $OUT/pmc_1 -o p --output-format csv -- python $ROOT/tools/run_stft_only.py $NFFT $NCLIP $TT 3 > $OUT/pmc_1.log 2>&1
python $ROOT/tools/pmc_summary.py $OUT stft_fwd
