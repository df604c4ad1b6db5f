#!/usr/bin/env python3
"""Turns a rocprofv3 (rocpd sqlite) result into the text summaries committed under profiles/.
    python tools/rocprof_summary.py <results.db> <out.txt> "<command line that was profiled>" [steps_total]
"""
import sqlite3
import sys


def main():
    db, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute('select name,total_calls,total_duration,average,percentage from top_kernels'))
    tot = sum(r[2] for r in rows)
    with open(out, 'w') as f:
        f.write('# %s\n# durations in microseconds; %d bench steps in the trace (warm-up included); '
                'total kernel time per step = %.1f us\n' % (cmd, steps, tot / steps))
        f.write('%-112s %8s %14s %12s %12s %7s\n' % ('kernel', 'calls', 'total_us', 'avg_us', 'us_per_step', 'pct'))
        for r in rows:
            f.write('%-112s %8d %14.1f %12.2f %12.1f %7.2f\n' % (r[0][:112], r[1], r[2], r[3], r[2] / steps, r[4]))
    print('wrote', out, 'kernel us/step', tot / steps)


if __name__ == '__main__':
    main()
