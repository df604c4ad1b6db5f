#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for v in "GHN3_FRESH_PREFETCH_THREAD=1" "GHN3_FRESH_PREFETCH_THREAD=0" "GHN3_FRESH_PREFETCH_THREAD=1" "GHN3_FRESH_PREFETCH_THREAD=0 GHN3_LOADER_WORKERS=4"; do
env $v timeout 900 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], 'fresh', d['fresh_graph_ms_per_step'], d['fresh_graph_host_ms'], 'gpu', d['fresh_graph_gpu_ms'])"
done
