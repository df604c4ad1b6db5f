#!/bin/bash
# scratch script for one-off GPU experiments (rewritten per experiment; see profiles/README.md for kept results)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_networks.py -m gpu -x -q 2>&1 | grep -E "passed|failed"
timeout 900 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'fresh', d['fresh_graph_ms_per_step'], d['fresh_graph_host_ms'], 'gpu', d['fresh_graph_gpu_ms'])"
