for v in 1 2 3; do
  python tools/r04/run_leg.py config4 2>&1 | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/config4 /"
done
timeout 300 python -m pytest tests/test_gpu_modules.py -x -q 2>&1 | tail -2
