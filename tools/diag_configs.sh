#!/bin/bash
# scratch script for one-off GPU experiments (rewritten per experiment; see profiles/README.md for kept results)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python bench.py --steps 3000 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{"metric' | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('3000 steps', d['ms_per_step'], d['roofline']['frac'])"
timeout 900 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --graphs-per-gpu 2 2>/dev/null | grep '^{"metric' | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=2', d['ms_per_step'], d['value']/1e9)"
timeout 900 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --model ghn3lm8 --nodes 200 2>/dev/null | grep '^{"metric' | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lm8 200 nodes', d['ms_per_step'], d['value']/1e9)"
timeout 900 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --model ghn3tm8 --nodes 128 --compute bf16 2>/dev/null | grep '^{"metric' | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tm8 128 nodes bf16', d['ms_per_step'], d['value']/1e9)"
timeout 900 python examples/train_synthetic.py --steps 30 2>&1 | tail -2
