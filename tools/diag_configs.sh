#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for v in "GHN3_X3_WIDE_SLICE=192" "GHN3_X3_WIDE_SLICE=128" "GHN3_X3_WIDE_SLICE=64" "GHN3_X3_WIDE_SLICE=384" "GHN3_X3_WIDE_SLICE=192"; do
env GHN3_X3_PLAN=D $v timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], 'fwd', d['forward']['ms'])"
done
