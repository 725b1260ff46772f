# A/B: hifi_gan_v1 config-3 step: a stage's resblocks on parallel streams (PSND_HIFIGAN_BRANCHES), the parameter-side backward launches on a side stream (PSND_BRANCH_PARAM_GRADS)
export PSND_LAB=1   # the host side reads its A/B switches only in a lab environment (pytorch_sound_amd/_switches.py)
for v in "1 1" "1 0" "1 1" "1 0"; do
  set -- $v
  PSND_HIFIGAN_BRANCHES=$1 PSND_BRANCH_PARAM_GRADS=$2 python tools/r04/run_leg.py config3 2>&1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/config3 branches=$1 param_side=$2 /"
done
for v in 1 0 1 0; do
  PSND_BRANCH_PARAM_GRADS=$v python bench.py --steps 40 --warmup 10 --cpu-seconds 0 --no-legs 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/config2 param_side=$v /"
done
timeout 500 python -m pytest tests/test_gpu_config3.py tests/test_gpu_hifigan.py tests/test_gpu_conv.py tests/test_gpu_trainer_graph.py -x -q 2>&1 | tail -3
# measured: side branches on high-priority streams (torch.cuda.Stream(priority=-1)): config-3 step 3.02-3.09 -> 5.93-5.97 ms; not kept
