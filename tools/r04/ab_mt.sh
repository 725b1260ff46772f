cd /tmp && export TMPDIR=/tmp
export PSND_LAB=1   # the host side reads its A/B switches only in a lab environment (pytorch_sound_amd/_switches.py)
R=$GRAFT_REPO_ROOT
for v in base mt2; do
  if [ $v = mt2 ]; then export PSND_CONV_MT=2; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$v -- python $R/bench.py --steps 40 --warmup 10 --cpu-seconds 0 --no-legs > $R/gpurun_out/ab_$v.log 2>&1
  python $R/tools/kstats.py /tmp/p_$v 30 $R/gpurun_out/ab_$v.txt > /dev/null
done
