# A/B: config-4 block: the 1x1 projections' parameter-side backward on the parameter stream (PSND_BRANCH_PARAM_GRADS), the two attention gradient kernels on two streams (PSND_ATTN_BWD_TWO_STREAMS)
export PSND_LAB=1   # the host side reads its A/B switches only in a lab environment (pytorch_sound_amd/_switches.py)
for v in "1 1" "1 0" "1 1" "1 0" "0 0"; do
  set -- $v
  PSND_BRANCH_PARAM_GRADS=$1 PSND_ATTN_BWD_TWO_STREAMS=$2 python tools/r04/run_leg.py config4 2>&1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/config4 param_side=$1 attn_two_streams=$2 /"
done
timeout 500 python -m pytest tests/test_gpu_modules.py tests/test_gpu_no_library_paths.py -x -q 2>&1 | tail -3
