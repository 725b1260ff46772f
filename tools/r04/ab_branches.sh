# A/B: hifi_gan_v1 config-3 step: a stage's resblocks on parallel streams (PSND_HIFIGAN_BRANCHES), the upsamplers' parameter-side backward on a side stream (PSND_BRANCH_PARAM_GRADS)
for v in "1 1" "1 0" "1 1" "1 0" "0 0"; do
  set -- $v
  PSND_HIFIGAN_BRANCHES=$1 PSND_BRANCH_PARAM_GRADS=$2 python tools/r04/run_leg.py config3 2>&1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/branches=$1 param_side=$2 /"
done
timeout 300 python -m pytest tests/test_gpu_config3.py tests/test_gpu_hifigan.py -x -q 2>&1 | tail -3
