"""Kernel timeline of the LAST step in a rocprofv3 database (start / duration / queue of every kernel from the last
graph_prologue_kernel on; [steps]: the last `steps` steps; [skip]: leave out that many steps at the end first -- bench.py ends
with a serialised pass of max(3, min(10, steps)) untimed steps):  python3 tools/rocprof_timeline.py <results.db> <out.csv> [steps] [skip]"""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
n_back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n_skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cur = sqlite3.connect(db).cursor()
views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
if 'kernels' not in views:
    print('no kernels view', views)
    sys.exit(0)
cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
qcol = 'queue_id' if 'queue_id' in cols else ('queue' if 'queue' in cols else None)
rows = list(cur.execute('select start, end, %s, name from kernels order by start' % (qcol or '0')))
starts = [i for i, r in enumerate(rows) if 'graph_prologue' in r[3]]
i1 = starts[-n_skip] if (starts and 0 < n_skip < len(starts)) else len(rows)
starts = [i for i in starts if i < i1]
i0 = starts[-min(n_back, len(starts))] if starts else 0
t0 = rows[i0][0]
qs = {}
with open(out, 'w') as f:
    f.write('start_us,dur_us,queue,kernel\n')
    for st, en, q, name in rows[i0:i1]:
        qs.setdefault(q, len(qs) + 1)
        f.write('%.2f,%.2f,%d,%s\n' % ((st - t0) / 1e3, (en - st) / 1e3, qs[q], name.replace(',', ';')[:60]))
print('timeline rows', i1 - i0)
