#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 300 python tests/attn_bench.py > gpurun_out/r02w_attn3.txt 2>&1; tail -4 gpurun_out/r02w_attn3.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "attention or attn or head or tiny or 1100" 2>&1 | tail -3
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{"metric' | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])"
