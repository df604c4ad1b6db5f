#!/bin/bash
# rocprofv3 kernel-trace summary of the bench workload (run on the GPU box through gpurun):
#   bash tools/profile_bench.sh <tag> [compute]      e.g.  bash tools/profile_bench.sh r01f f16
# Writes gpurun_out/<tag>_rocprof_kernel_stats_xl_<compute>.txt, <tag>_bench_under_rocprof_xl_<compute>.json and
# <tag>_kernel_timeline_last_step_xl_<compute>.csv (start / duration / queue of every kernel of the last step).
TAG=${1:-prof}; CT=${2:-f16}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS=5; WARM=2
rm -rf /tmp/prof_$TAG
CMD="python3 bench.py --compute $CT --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extras"
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- $CMD > gpurun_out/${TAG}_bench_under_rocprof_xl_$CT.json 2> /tmp/prof_err.log
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
if [ -z "$DB" ]; then echo "no db"; tail -5 /tmp/prof_err.log; exit 1; fi
python3 tools/rocprof_summary.py "$DB" gpurun_out/${TAG}_rocprof_kernel_stats_xl_$CT.txt "rocprofv3 --kernel-trace --stats -- $CMD (ghn3xlm16, one 256-node graph, side stream on)" $((STEPS+WARM))
python3 - "$DB" gpurun_out/${TAG}_kernel_timeline_last_step_xl_$CT.csv $((STEPS+WARM)) <<'PY'
import sqlite3, sys
db, out, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
cur = sqlite3.connect(db).cursor()
views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
view = 'kernels' if 'kernels' in views else None
if view is None:
    print('no kernels view', views); sys.exit(0)
cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
qcol = 'queue_id' if 'queue_id' in cols else ('queue' if 'queue' in cols else None)
rows = list(cur.execute('select start, end, %s, name from kernels order by start' % (qcol or '0')))
# the last step starts with the first graph_prologue_kernel of the final 1/steps of the run
starts = [i for i, r in enumerate(rows) if 'graph_prologue' in r[3]]
i0 = starts[-1] if starts else 0
t0 = rows[i0][0]
qs = {}
with open(out, 'w') as f:
    f.write('start_us,dur_us,queue,kernel\n')
    for st, en, q, name in rows[i0:]:
        qs.setdefault(q, len(qs) + 1)
        f.write('%.2f,%.2f,%d,%s\n' % ((st - t0) / 1e3, (en - st) / 1e3, qs[q], name.replace(',', ';')[:60]))
print('timeline rows', len(rows) - i0)
PY
head -30 gpurun_out/${TAG}_rocprof_kernel_stats_xl_$CT.txt | cut -c1-60,110-200
tail -1 gpurun_out/${TAG}_bench_under_rocprof_xl_$CT.json | cut -c1-300
