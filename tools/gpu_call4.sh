#!/bin/bash
# round 6: what the training loop of examples/train_ghn_ddp.py spends its GPU time on (kernel names, counts, busy fraction)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06i
export TMPDIR=/tmp MIOPEN_FIND_MODE=3
rm -rf /tmp/prof_train; timeout 600 python3 examples/train_ghn_ddp.py --steps 23 > gpurun_out/r06i/train_pass1.log 2>&1; tail -2 gpurun_out/r06i/train_pass1.log
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o r -- python3 examples/train_ghn_ddp.py --steps 23 > gpurun_out/r06i/train.log 2>&1
tail -3 gpurun_out/r06i/train.log
DB=$(find /tmp/prof_train -name "*.db" | head -1)
python3 - "$DB" <<'PY' > gpurun_out/r06i/train_kernels.txt
import sqlite3, sys
from collections import defaultdict
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = 'kernels' if 'kernels' in tabs else [t for t in tabs if 'kernel' in t.lower()][0]
cols = [r[1] for r in cur.execute('pragma table_info(%s)' % view)]
name = 'name' if 'name' in cols else 'kernel_name'
rows = cur.execute('select %s, start, end from %s order by start' % (name, view)).fetchall()
t0, t1 = rows[0][1], rows[-1][2]
agg = defaultdict(lambda: [0, 0.0])
busy, last = 0.0, t0
for n, s, e in rows:
    k = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:90]
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
    if e > last:
        busy += (e - max(s, last)) / 1e3; last = e
print('# rocprofv3 --kernel-trace -- python3 examples/train_ghn_ddp.py --steps 23 (ghn3tm8, meta-batch 8, 64 images): %d kernel launches, span %.1f ms, GPU busy %.1f ms (%.0f %%)'
      % (len(rows), (t1 - t0) / 1e6, busy / 1e3, 100 * busy * 1e3 / (t1 - t0)))
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%-92s calls %7d total_ms %9.1f avg_us %8.1f' % (k, c, us / 1e3, us / c))
PY
head -50 gpurun_out/r06i/train_kernels.txt | cut -c1-170
