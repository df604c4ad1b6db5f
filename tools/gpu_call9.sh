#!/bin/bash
# round 6, call 9: flat gradient buffer zeroed beside the forward (GHN3_GRAD_PREZERO) -- tests, bench A/B on one box
set -u
mkdir -p gpurun_out/r06y
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r06y/tests.txt
for rep in 1 2; do
for pz in 0 1; do
  GHN3_GRAD_PREZERO=$pz timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r06y/bench_prezero${pz}_${rep}.json 2> gpurun_out/r06y/bench_prezero${pz}_${rep}.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r06y/bench_prezero${pz}_${rep}.json'))
print('prezero=$pz pass=$rep ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'fwd', d['roofline'].get('forward_ms'), d.get('forward',{}).get('ms'))
PY
done
done 2>&1 | tee gpurun_out/r06y/ab.txt
