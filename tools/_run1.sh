cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -15 > gpurun_out/t16.log
for ns in 0 1; do
for c in f16 f32; do
GHN3_NO_SIDE_STREAM=$ns timeout 300 python bench.py --compute $c --no-cpu-baseline --steps 10 --warmup 3 --profile-ops > gpurun_out/b_${c}_$ns.log 2>&1
done; done
cat gpurun_out/t16.log; python - <<'PY'
import json
for f in ('gpurun_out/b_f16_0.log', 'gpurun_out/b_f16_1.log', 'gpurun_out/b_f32_0.log', 'gpurun_out/b_f32_1.log'):
  for l in open(f):
    if l.startswith('{'):
        d = json.loads(l)
        print(f, d['ms_per_step'], d['value'], d['roofline']['achieved'])
        print({k: (v['ms_per_step'], v.get('tflops')) for k, v in d['roofline']['kernels'].items()})
PY
