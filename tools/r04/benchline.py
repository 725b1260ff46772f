"""pick the JSON line out of a bench.py run's output and print a few numbers: python bench.py ... | python tools/r04/benchline.py tag"""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ''
for ln in sys.stdin.read().splitlines():
    if ln.startswith('{"metric"'):
        d = json.loads(ln)
        print('[%s] ms/step %.4f  value %.0f  median %.4f  min %.4f' % (tag, d['ms_per_step'], d['value'], d['timing']['median'], d['timing']['min']))
