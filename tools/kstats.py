#!/usr/bin/env python
"""print the top rows of a rocprofv3 `--kernel-trace --stats --output-format csv` run:  tools/kstats.py <dir> [rows] [outfile]"""
import csv, glob, sys
d = sys.argv[1]; rows = int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)
out = []
if not f:
    out.append('no kernel_stats.csv under ' + d)
else:
    rs = list(csv.DictReader(open(f[0])))
    tot = sum(float(r['TotalDurationNs']) for r in rs)
    out.append('%-100s %7s %12s %10s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', '%'))
    for r in rs[:rows]:
        out.append('%-100s %7s %12.1f %10.2f %7.2f' % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
    out.append('TOTAL kernel time %.1f us over %d dispatches, %d distinct kernels' % (tot / 1e3, sum(int(r['Calls']) for r in rs), len(rs)))
txt = '\n'.join(out)
print(txt)
if len(sys.argv) > 3:
    open(sys.argv[3], 'w').write(txt + '\n')
