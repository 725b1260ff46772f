#!/bin/bash
# end-of-round profile set after the graph-branch work (no counter passes: the STFT kernels did not change): kernel stats + timelines of the
# bench step, config 3, config 4 -> gpurun_out/r04/
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python $ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-legs > $OUT/bench_prof.log 2>&1
python $ROOT/tools/kstats.py /tmp/p_bench 40 $OUT/r04_bench_step_kernel_stats.txt > /dev/null
python $ROOT/tools/step_timeline.py /tmp/p_bench 30 $OUT/r04_step_timeline.txt > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3 -- python $ROOT/tools/r04/run_leg.py config3 > $OUT/c3.log 2>&1
python $ROOT/tools/kstats.py /tmp/p_c3 70 $OUT/r04_config3_kernel_stats.txt > /dev/null
python $ROOT/tools/step_timeline.py /tmp/p_c3 30 $OUT/r04_config3_timeline.txt > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python $ROOT/tools/r04/run_leg.py config4 > $OUT/c4.log 2>&1
python $ROOT/tools/kstats.py /tmp/p_c4 40 $OUT/r04_config4_block_bf16_kernel_stats.txt > /dev/null
python $ROOT/tools/step_timeline.py /tmp/p_c4 20 $OUT/r04_config4_timeline.txt > /dev/null
grep -h -o '"ms_per_step": [0-9.]*' $OUT/c3.log $OUT/c4.log
cd $ROOT
python bench.py > $OUT/bench_full.log 2> $OUT/bench_full.err
tail -c 600 $OUT/bench_full.log
