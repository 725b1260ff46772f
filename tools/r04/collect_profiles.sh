#!/bin/bash
# round-4 profile set (run on the GPU box through gpurun): kernel stats + step timeline of the bench step, config-3 / config-4 kernel stats,
# PMC passes of the four STFT launches the bench line quotes -> gpurun_out/r04/
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -- python $ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-legs > $OUT/bench_prof.log 2>&1
python $ROOT/tools/kstats.py /tmp/p_bench 40 $OUT/r04_bench_step_kernel_stats.txt > /dev/null
python $ROOT/tools/step_timeline.py /tmp/p_bench 30 $OUT/r04_step_timeline.txt > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c3 -- python $ROOT/tools/r04/run_leg.py config3 > $OUT/c3.log 2>&1
python $ROOT/tools/kstats.py /tmp/p_c3 70 $OUT/r04_config3_kernel_stats.txt > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python $ROOT/tools/r04/run_leg.py config4 > $OUT/c4.log 2>&1
python $ROOT/tools/kstats.py /tmp/p_c4 40 $OUT/r04_config4_block_bf16_kernel_stats.txt > /dev/null
cd $ROOT
bash tools/pmc_stft.sh r04/pmc1024 1024 1024 44100 > /dev/null 2>&1
bash tools/pmc_stft.sh r04/pmc4096 4096 32 1323000 > /dev/null 2>&1
bash tools/pmc_stft.sh r04/pmcnfk1024 1024 1024 44100 RUNNER=tools/r04/run_nfk_only.py > /dev/null 2>&1
bash tools/pmc_stft.sh r04/pmcnfk4096 4096 32 1323000 RUNNER=tools/r04/run_nfk_only.py > /dev/null 2>&1
python tools/pmc_to_json.py n1024=gpurun_out/r04/pmc1024 n4096=gpurun_out/r04/pmc4096 nfk1024=gpurun_out/r04/pmcnfk1024 nfk4096=gpurun_out/r04/pmcnfk4096 > $OUT/pmc_to_json.log 2>&1
cp profiles/stft_pmc.json $OUT/stft_pmc.json 2>/dev/null
# keep the merge small: drop the raw counter dumps, keep the summaries
for d in pmc1024 pmc4096 pmcnfk1024 pmcnfk4096; do rm -rf $OUT/$d/pmc_*; done
grep -h ms_per_step $OUT/c3.log $OUT/c4.log | cut -c1-200
