#!/usr/bin/env python
"""how much of the kernel time in a rocprofv3 kernel-trace CSV ran concurrently with another kernel (parallel graph branches / streams)"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:50], r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in csv.DictReader(open(f))]
rows.sort()
tot = sum(e - s for s, e, _, _ in rows)
ev = sorted([(s, 1) for s, e, _, _ in rows] + [(e, -1) for s, e, _, _ in rows])
busy = 0; par = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: par += t - last
    depth += d; last = t
print('kernels %d, sum of durations %.1f ms, wall with >=1 kernel %.1f ms, with >=2 kernels %.1f ms' % (len(rows), tot / 1e6, busy / 1e6, par / 1e6))
print('queues/streams:', sorted(set(r[3] for r in rows)))
