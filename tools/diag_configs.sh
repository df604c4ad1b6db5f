cd "${GRAFT_REPO_ROOT:-.}"
for args in "ghn3xlm16 f16 40" "ghn3lm8 f16 25 40"; do
  timeout 600 python tests/gpu_diag_configs.py $args 2>&1 | grep -v amdgpu.ids | head -7 | cut -c1-150
done
for F in 1 0; do
echo "== D2_FIX $F"; GHN3_D2_FIX=$F python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', round(d['ms_per_step'],3), {k: v['ms_per_step'] for k, v in d['roofline']['kernels'].items() if k in ('w0_fwd','fc_fwd','w2_fwd')})"
done
