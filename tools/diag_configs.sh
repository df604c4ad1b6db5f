#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "distributed_data_parallel" 2>&1 | tail -12
O=gpurun_out/r02t_force_ddp.txt; : > $O
for a in "--force-ddp" "--force-ddp --grad-allreduce f32" "--force-ddp --grad-allreduce f32-serial"; do
  echo "== $a" >> $O
  timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras $a 2>>gpurun_out/r02t.err | grep '^{"metric' | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('grad_allreduce'))" >> $O
done
cat $O
