cd "${GRAFT_REPO_ROOT:-.}"
for i in 1 2; do
for CS in 1 0; do
  echo "== copy stream $CS"
  GHN3_COPY_STREAM=$CS python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras --profile-ops 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', round(d['ms_per_step'],3), 'phases', d.get('phase_ms'))"
done
done
