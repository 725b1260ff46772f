# A/B: config-4 block with the 1x1 projections' parameter-side backward on the parameter stream (PSND_BRANCH_PARAM_GRADS)
for v in 1 0 1 0; do
  PSND_BRANCH_PARAM_GRADS=$v python tools/r04/run_leg.py config4 2>&1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/config4 param_side=$v /"
done
timeout 500 python -m pytest tests/test_gpu_modules.py tests/test_gpu_no_library_paths.py -x -q 2>&1 | tail -3
