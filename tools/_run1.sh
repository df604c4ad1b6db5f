cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "split_backward" 2>&1 | tail -3 > gpurun_out/t16.log
cat gpurun_out/t16.log
for m in f32-serial f32 bf16; do
timeout 300 python bench.py --force-ddp --grad-allreduce $m --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/b_ddp_$m.log 2>&1
grep "^{" gpurun_out/b_ddp_$m.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'], d['config'].get('grad_allreduce'))"
done
