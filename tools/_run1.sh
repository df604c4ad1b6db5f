cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "tiny_backward or ghn3tm8 or split_backward or properties" 2>&1 | grep -E "passed|failed|rror|assert" | head
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --profile-ops > gpurun_out/b_f16_0.log 2>&1
python - <<'PY'
import json
for f in ('gpurun_out/b_f16_0.log',):
  for l in open(f):
    if l.startswith('{'):
        d = json.loads(l)
        print(f, d['ms_per_step'], d['value'], d['roofline']['achieved'], d['phase_ms'])
        print({k: (v['ms_per_step'], v.get('tflops'), v['launch_groups_per_step']) for k, v in d['roofline']['kernels'].items() if k.startswith('w2')})
PY
done
