#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
O=gpurun_out/r02q_group.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>>gpurun_out/r02q.err | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print(d['ms_per_step'], {n:(v['ms_per_step']) for n,v in k.items() if n.startswith('w2')})" >> $O; }
run GHN3_SIDE_GROUP=1
run GHN3_SIDE_GROUP=2
run GHN3_SIDE_GROUP=4
run GHN3_SIDE_GROUP=8
run GHN3_SIDE_GROUP=1
run GHN3_SIDE_GROUP=4
run GHN3_SIDE_GROUP=24
cat $O
