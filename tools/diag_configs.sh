cd "${GRAFT_REPO_ROOT:-.}"
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', round(d['ms_per_step'],3), 'fresh', d.get('fresh_graph_ms_per_step'), d.get('fresh_graph_host_ms'))"
