# round-5 baseline on this round's box: GPU suite, smoke, driver-flag bench line
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r05/baseline_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r05/baseline_tests.log
python bench.py --steps 20 --warmup 5 2>gpurun_out/r05/baseline_bench.err | tail -1 > gpurun_out/r05/baseline_bench.json
cat gpurun_out/r05/baseline_tests.log
