#!/bin/bash
# Is the training loop GPU-bound?  rocprofv3 kernel trace of 24 Trainer.update steps on the example's architecture stream
# (in-process, no worker pool -- a spawn under the profiler is an exec after GPU initialisation --: tools/diag/stock_layer_census.py),
# native layers on; prints the kernel table, GPU busy time per step and the dense weight gradient's calls by grid.
#   bash tools/train_loop_trace.sh   ->  gpurun_out/r06y/train_loop_rocprof_kernel_stats.txt
set -u
mkdir -p gpurun_out/r06y
export TMPDIR=/tmp
cd /tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_loop
CENSUS_STEPS=24 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_loop -o r -- python3 $ROOT/tools/diag/stock_layer_census.py > $ROOT/gpurun_out/r06y/census_under_rocprof.txt 2> /tmp/prof_err.log
DB=$(find /tmp/prof_loop -name "*.db" | head -1)
if [ -z "$DB" ]; then echo "no db"; tail -5 /tmp/prof_err.log; exit 1; fi
cd $ROOT
python3 tools/rocprof_summary.py "$DB" gpurun_out/r06y/train_loop_rocprof_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- python3 tools/diag/stock_layer_census.py (24 steps of Trainer.update, ghn3tm8, meta-batch 8, 64 images of 32 x 32; the architecture stream of examples/train_ghn_ddp.py)" 24
head -45 gpurun_out/r06y/train_loop_rocprof_kernel_stats.txt | cut -c1-100,113-170
python3 - <<PY
import sqlite3
cur = sqlite3.connect("$DB").cursor()
ks = list(cur.execute('select start, end from kernels order by start'))
n = len(ks)
# busy time = union of kernel intervals; wall = last end - first start
busy, cur_s, cur_e = 0, None, None
for s, e in ks:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = ks[-1][1] - ks[0][0]
print('kernels %d, per step %.0f; GPU busy (union) %.1f ms per step; wall %.1f ms per step' % (n, n / 24, busy / 24e6, wall / 24e6))
PY
python3 - <<PY
import sqlite3
cur = sqlite3.connect("$DB").cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
want = [c for c in ('grid_x','grid_y','grid_z','workgroup_x','grid_size_x','grid_size_y','grid_size_z','workgroup_size_x','lds_size','lds_block_size') if c in cols]
q = 'select (end-start)/1000.0 as us, %s from kernels where name like "%%tnet_conv_wgrad%%" order by us desc' % ','.join(want)
rows = list(cur.execute(q))
import collections
tot = sum(r[0] for r in rows)
print('conv_wgrad calls', len(rows), 'total us', tot)
acc = 0
for r in rows[:12]: print(r)
b = collections.Counter()
for r in rows: b[(r[1:], )] += r[0]
for k, v in sorted(b.items(), key=lambda kv: -kv[1])[:25]: print(round(v), k, sum(1 for r in rows if (r[1:],) == k))
PY
