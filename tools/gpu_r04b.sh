set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
python bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline --profile-ops > gpurun_out/r04b/bench_quick.json 2> gpurun_out/r04b/bench_quick.err
tail -c 3000 gpurun_out/r04b/bench_quick.json
GHN3_X3S=0 python bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r04b/bench_quick_old.json 2>> gpurun_out/r04b/bench_quick.err
python - <<'PY'
import json
for f in ('bench_quick', 'bench_quick_old'):
    d = json.loads(open('gpurun_out/r04b/%s.json' % f).read().strip().split('\n')[-1])
    print(f, d['ms_per_step'], d['roofline']['frac'], d.get('phase_ms'))
PY
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -15 > gpurun_out/r04b/tests.txt
cat gpurun_out/r04b/tests.txt
