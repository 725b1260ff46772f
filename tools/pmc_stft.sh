#!/bin/sh
# PMC passes over psnd_stft_fwd (n=1024, N clips): tools/pmc_stft.sh <outdir> <N> "<set1>" "<set2>" ...
# each set = space-separated counter names collected in ONE rocprofv3 pass (--kernel-trace only, as gpurun requires)
out=$1; N=$2; shift 2
cd /tmp; export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $root/$out -o p$i -- python $root/tools/run_stft_only.py ${NFFT:-1024} $N ${TLEN:-44100} 3 > $root/$out/pass$i.log 2>&1
  mkdir -p $root/$out/pmc_$i; mv $root/$out/p${i}_*.csv $root/$out/pmc_$i/ 2>/dev/null
done
python $root/tools/pmc_summary.py $root/$out stft_fwd
