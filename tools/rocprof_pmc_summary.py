#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counters per kernel name from a rocpd sqlite database.
    python tools/rocprof_pmc_summary.py <results.db> <label> [substring filter]
Prints one line per kernel: calls, summed duration, summed counter values."""
import sqlite3
import sys
from collections import defaultdict


def main():
    db, label = sys.argv[1], sys.argv[2]
    flt = sys.argv[3] if len(sys.argv) > 3 else ''
    cur = sqlite3.connect(db).cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
    view = 'counters_collection' if 'counters_collection' in views else None
    if view is None:
        print('no counters_collection view; have', views)
        return
    cols = [r[1] for r in cur.execute('pragma table_info(%s)' % view)]
    name_col = 'kernel_name' if 'kernel_name' in cols else 'name'
    rows = cur.execute('select %s, dispatch_id, counter_name, value, start, end from %s' % (name_col, view)).fetchall()
    agg = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(dict)
    for name, did, cname, val, st, en in rows:
        key = name.replace('(anonymous namespace)::', '').split('(')[0][:70]
        agg[key][cname] += val
        disp[key][did] = (en - st)
    for key, d in sorted(agg.items(), key=lambda kv: -sum(disp[kv[0]].values())):
        if flt and flt not in key:
            continue
        dur = sum(disp[key].values()) / 1e6
        print('%s | %s calls=%d dur_ms=%.2f %s' % (label, key, len(disp[key]), dur,
                                                   ' '.join('%s=%.4g' % (k, v) for k, v in sorted(d.items()))))


if __name__ == '__main__':
    main()
