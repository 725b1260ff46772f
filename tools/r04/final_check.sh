# end-of-round check on the GPU box: the whole -m gpu suite, smoke(), the bench line with the driver's flags
timeout 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_final.log
python - <<'P'
import json
d = json.loads(open('gpurun_out/bench_final.log').read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config3_step'].get('ms_per_step'), d['config4_step'].get('ms_per_step'))
P
