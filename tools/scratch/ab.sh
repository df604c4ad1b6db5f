python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | grep -E "^E |passed|failed" | head -12
for i in 1 2; do
for v in 0 1; do
GHN3_SPLIT_K2=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-ops 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d.get('phase_ms'); print('split=$v', round(d['ms_per_step'],3), p['forward'], p['backward'])"
done; done
bash tools/profile_bench.sh r01k_split2 f16 >/dev/null 2>&1
