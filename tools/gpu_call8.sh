#!/bin/bash
# round 6, call 8: ghn3_run's store of resolved problem tables -- tests, then bench A/B (GHN3_RUN_CACHE=0 / 1) on one box
set -u
mkdir -p gpurun_out/r06x
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -x -q -m gpu -k "reuse or precompiled or tiny_forward or tiny_backward or split_backward or fused_adamw" 2>&1 | tail -5 | tee gpurun_out/r06x/tests.txt
for rep in 1 2; do
for rc in 0 1; do
  GHN3_RUN_CACHE=$rc timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r06x/bench_cache${rc}_${rep}.json 2> gpurun_out/r06x/bench_cache${rc}_${rep}.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r06x/bench_cache${rc}_${rep}.json'))
print('run_cache=$rc pass=$rep ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'fwd', d['roofline'].get('forward_ms'))
PY
done
done 2>&1 | tee gpurun_out/r06x/ab.txt
for cfg in resnet50-xl tm8-128; do
for rc in 0 1; do
  GHN3_RUN_CACHE=$rc timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-extras > gpurun_out/r06x/cfg_${cfg}_cache${rc}.json 2> gpurun_out/r06x/cfg_${cfg}_cache${rc}.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r06x/cfg_${cfg}_cache${rc}.json'))
print('$cfg run_cache=$rc ms_per_step', d['ms_per_step'])
PY
done
done 2>&1 | tee -a gpurun_out/r06x/ab.txt
