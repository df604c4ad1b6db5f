cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cast16 or tiny_forward" 2>&1 | grep -E "passed|failed|rror" 
for t in 0 2048 1024 512 256; do
GHN3_W2CAST_CAP=$t timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --profile-ops > gpurun_out/b_f16_t$t.log 2>&1
python - <<PY
import json
for l in open('gpurun_out/b_f16_t$t.log'):
    if l.startswith('{'):
        d = json.loads(l)
        print($t, d['ms_per_step'], d['phase_ms']['forward'], d['phase_ms']['backward'])
PY
done
