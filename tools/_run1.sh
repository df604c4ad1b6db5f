cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in 0 6 4; do
GHN3_SIDE_CU_MASK=$m timeout 300 python bench.py --compute f16 --no-cpu-baseline --steps 10 --warmup 3 --profile-ops > gpurun_out/b_f16_m$m.log 2>&1
done
python - <<'PY'
import json
for m in (0, 6, 4):
  f = 'gpurun_out/b_f16_m%d.log' % m
  for l in open(f):
    if l.startswith('{'):
        d = json.loads(l)
        print(f, d['ms_per_step'], d['value'], d['roofline']['achieved'], d['phase_ms'])
    elif 'rror' in l: print(l[:200])
PY
