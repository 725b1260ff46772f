#!/bin/bash
# round-6 profile set (on the GPU box through gpurun): kernel stats + step timeline of the bench step, config-3 / config-4 / drop-in kernel
# stats, the kernel trace of the two (N, F, K) STFT launches the bench line quotes -> gpurun_out/r06/
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python $ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-legs > $OUT/bench_prof.log 2>&1
python $ROOT/tools/kstats.py /tmp/p_bench 40 $OUT/r06_bench_step_kernel_stats.txt > /dev/null
python $ROOT/tools/step_timeline.py /tmp/p_bench 30 $OUT/r06_step_timeline.txt > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3 -- python $ROOT/bench.py --leg config3 > $OUT/c3.log 2>&1
python $ROOT/tools/kstats.py /tmp/p_c3 70 $OUT/r06_config3_kernel_stats.txt > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python $ROOT/bench.py --leg config4 > $OUT/c4.log 2>&1
python $ROOT/tools/kstats.py /tmp/p_c4 40 $OUT/r06_config4_block_bf16_kernel_stats.txt > /dev/null
python $ROOT/tools/step_timeline.py /tmp/p_c4 30 $OUT/r06_config4_timeline.txt > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -- python $ROOT/bench.py --config 5 --steps 20 --warmup 5 --settle 5 > $OUT/c5.log 2>&1
python $ROOT/tools/kstats.py /tmp/p_c5 10 $OUT/r06_config5_kernel_stats.txt > /dev/null
grep -h -o '"ms_per_step": [0-9.]*' $OUT/c3.log $OUT/c4.log $OUT/c5.log
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_di -- python $ROOT/bench.py --leg dropin > $OUT/dropin.log 2>&1
python $ROOT/tools/kstats.py /tmp/p_di 60 $OUT/r06_dropin_kernel_stats.txt > /dev/null
grep -h -o '"ms_per_step": [0-9.]*' $OUT/dropin.log
