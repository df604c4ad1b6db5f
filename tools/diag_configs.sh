#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "16bit_operand" 2>&1 | tail -5
O=gpurun_out/r02ze_pin.txt; : > $O
run() { echo "== $*" >> $O; env "$@" timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{"metric' | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print(d['ms_per_step'], d['roofline']['frac'], {n:(v['ms_per_step']) for n,v in k.items() if n.startswith('w2')})" >> $O; }
run GHN3_DGRAD_PIN=0
run GHN3_DGRAD_PIN=1
run GHN3_DGRAD_PIN=0
run GHN3_DGRAD_PIN=1
cat $O
bash tools/pmc_profile.sh r02ze > /dev/null 2>&1; grep "FETCH_SIZE" gpurun_out/r02ze_pmc_xl_f16.txt | head -5
