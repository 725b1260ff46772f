#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel table
(name, calls, total/avg/min/max duration) - the same content as `--stats` CSV output."""
import sqlite3
import sys


def main(path, top=40, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by %s order by 3 desc" % (namecol, namecol)).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ['%-90s %7s %12s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%')]
    for r in rows[:top]:
        lines.append('%-90s %7d %12.1f %10.2f %10.2f %10.2f %6.2f' % (r[0][:90], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                                      r[5] / 1e3, 100.0 * r[2] / total))
    lines.append('TOTAL kernel time %.1f us over %d dispatches, %d distinct kernels' % (total / 1e3, sum(r[1] for r in rows), len(rows)))
    # the STFT forward kernel is launched on two workloads in bench.py (in-step: 32 clips; roofline_large: 1024 clips):
    # report them apart, split at 50 us
    for r in cur.execute("select %s, sum(end-start < 50000), avg(case when end-start < 50000 then end-start end), "
                         "sum(end-start >= 50000), avg(case when end-start >= 50000 then end-start end) from kernels "
                         "where %s like '%%stft_fwd_n1024%%' group by %s" % (namecol, namecol, namecol)).fetchall():
        lines.append('stft_fwd_n1024 split: %d launches < 50 us, avg %.2f us (in-step, 64 clips: mixture + reference of a batch) | %d launches >= 50 us, avg %.2f us (1024 clips)'
                     % (r[1] or 0, (r[2] or 0) / 1e3, r[3] or 0, (r[4] or 0) / 1e3))
    txt = '\n'.join(lines)
    print(txt)
    if out:
        open(out, 'w').write(txt + '\n')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40, sys.argv[3] if len(sys.argv) > 3 else None)
