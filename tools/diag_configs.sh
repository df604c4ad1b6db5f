#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
O=gpurun_out/r02u_wgrad25.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>>gpurun_out/r02q.err | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print(d['ms_per_step'], d['roofline']['frac'], {n:(v['ms_per_step']) for n,v in k.items() if n.startswith('w2')})" >> $O; }
run GHN3_WGRAD_TILE=0
run GHN3_WGRAD_TILE=25
run GHN3_WGRAD_TILE=25 GHN3_WGRAD_CAP=224
run GHN3_WGRAD_TILE=25 GHN3_WGRAD_CAP=192
run GHN3_WGRAD_TILE=0
run GHN3_WGRAD_TILE=25
cat $O
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
