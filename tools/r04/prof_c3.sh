ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3 -- python $ROOT/tools/r04/run_leg.py config3 > $OUT/c3.log 2>&1
python $ROOT/tools/kstats.py /tmp/p_c3 70 $OUT/r04_config3_kernel_stats.txt > /dev/null
python $ROOT/tools/step_timeline.py /tmp/p_c3 30 $OUT/r04_config3_timeline.txt > /dev/null
grep -h -o '"ms_per_step": [0-9.]*' $OUT/c3.log
