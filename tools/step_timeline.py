#!/usr/bin/env python
"""kernel sequence of ONE optimizer step from a rocprofv3 `--kernel-trace --output-format csv` run of bench.py: the dispatches between
two consecutive adam_kernel launches, in start order, with duration, the gap to the previous kernel's end and whether it overlapped.
tools/step_timeline.py <dir> [which adam launch, default 30] [outfile]"""
import csv, glob, sys
d = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
rs = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
ad = [i for i, r in enumerate(rs) if 'adam_kernel' in r['Kernel_Name']]
a, b = ad[which - 1], ad[which]
out = []
t0 = int(rs[a]['End_Timestamp']); prev_end = t0; busy = 0
out.append('%8s %8s %8s  %s' % ('start_us', 'dur_us', 'gap_us', 'kernel'))
for r in rs[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.append('%8.1f %8.2f %8.2f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r['Kernel_Name'][:110]))
    busy += e - s
    prev_end = max(prev_end, e)
out.append('step: %.1f us wall from the previous adam_kernel end to this one, %.1f us summed kernel time, %d dispatches' % (
    (prev_end - t0) / 1e3, busy / 1e3, b - a))
txt = '\n'.join(out)
print(txt)
if len(sys.argv) > 3:
    open(sys.argv[3], 'w').write(txt + '\n')
