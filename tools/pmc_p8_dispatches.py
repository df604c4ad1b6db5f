#!/usr/bin/env python3
"""Per-dispatch rocprofv3 --pmc counters of one kernel, grouped by the dispatch's position inside a step (the W2 forward and
the W2 dgrad are both `gemm_p8_kernel`: position 0 / 1 of every step).
    python tools/pmc_p8_dispatches.py <results.db> <kernel substring> <dispatches per step> [label]
Prints, per position: launches, mean duration (us) and the mean of every counter (FETCH_SIZE / WRITE_SIZE in KiB as reported;
on gfx950 FETCH_SIZE counts half of the bytes of wide loads -- MI355X_MICROARCH.md)."""
import sqlite3
import sys
from collections import defaultdict


def main():
    db, sub, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
    label = sys.argv[4] if len(sys.argv) > 4 else ''
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(counters_collection)')]
    name_col = 'kernel_name' if 'kernel_name' in cols else 'name'
    rows = cur.execute('select %s, dispatch_id, counter_name, value, start, end from counters_collection' % name_col).fetchall()
    disp = {}
    for name, did, cname, val, st, en in rows:
        if sub not in name:
            continue
        d = disp.setdefault(did, {'dur': (en - st) / 1e3, 'c': defaultdict(float)})
        d['c'][cname] += val
    ids = sorted(disp)
    for pos in range(per):
        sel = [disp[i] for k, i in enumerate(ids) if k % per == pos]
        if not sel:
            continue
        names = sorted(sel[0]['c'])
        print('%s %s pos %d: n=%d dur_us=%.1f %s' % (label, sub, pos, len(sel), sum(d['dur'] for d in sel) / len(sel),
              ' '.join('%s=%.5g' % (n, sum(d['c'][n] for d in sel) / len(sel)) for n in names)))


if __name__ == '__main__':
    main()
