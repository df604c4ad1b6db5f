#!/usr/bin/env python3
"""Turns a rocprofv3 (rocpd sqlite) result into the text summaries committed under profiles/.
    python tools/rocprof_summary.py <results.db> <out.txt> "<command line that was profiled>" [steps_total] [serialised_steps]
With `serialised_steps` (bench.py ends with that many untimed steps in profile mode 3: side ops on the chain's stream) the file
holds two more tables: the scheduled steps alone and the serialised pass alone (the durations `roofline.achieved` is built from).
"""
import sqlite3
import sys


def table(f, rows, steps):
    tot = sum(r[2] for r in rows)
    f.write('%-112s %8s %14s %12s %12s %7s\n' % ('kernel', 'calls', 'total_us', 'avg_us', 'us_per_step', 'pct'))
    for r in rows:
        f.write('%-112s %8d %14.1f %12.2f %12.1f %7.2f\n' % (r[0][:112], r[1], r[2], r[2] / r[1], r[2] / steps, 100.0 * r[2] / tot))
    return tot


def grouped(ks):
    acc = {}
    for st, en, name in ks:
        c = acc.setdefault(name, [0, 0.0])
        c[0] += 1
        c[1] += (en - st) / 1e3
    return sorted(((n, c[0], c[1]) for n, c in acc.items()), key=lambda r: -r[2])


def main():
    db, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    n_ser = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    cur = sqlite3.connect(db).cursor()
    rows = [(r[0], r[1], r[2]) for r in cur.execute('select name,total_calls,total_duration from top_kernels')]
    tot = sum(r[2] for r in rows)
    with open(out, 'w') as f:
        f.write('# %s\n# durations in microseconds; %d bench steps in the trace (warm-up included); '
                'total kernel time per step = %.1f us\n' % (cmd, steps, tot / steps))
        table(f, rows, steps)
        if n_ser > 0:
            ks = list(cur.execute('select start, end, name from kernels order by start'))
            starts = [i for i, k in enumerate(ks) if 'graph_prologue' in k[2]]
            if len(starts) > n_ser:
                first, cut = starts[0], starts[-n_ser]
                f.write('\n# the %d scheduled steps alone (warm-up + timed region: side stream on)\n' % (len(starts) - n_ser))
                table(f, grouped(ks[first:cut]), len(starts) - n_ser)
                f.write('\n# the %d serialised steps alone (bench.py\'s roofline pass, profile mode 3: every kernel alone on the chip)\n' % n_ser)
                table(f, grouped(ks[cut:]), n_ser)
    print('wrote', out, 'kernel us/step', tot / steps)


if __name__ == '__main__':
    main()
