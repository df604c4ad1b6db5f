#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06h
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cast16_and_16bit" 2>&1 | tail -3
sel='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], "ms", round(d["ms_per_step"],4), "frac", round(d["roofline"]["frac"],4), {k:v["ms_per_step"] for k,v in d["roofline"]["kernels"].items()})'
for cfg in resnet50-xl vit-xl resnet18-tm8; do
  for mr in 160 8; do
    GHN3_P8_MIN_ROWS=$mr python bench.py --config $cfg --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r06h/${cfg}_min$mr.json 2> gpurun_out/r06h/${cfg}_min$mr.err
    python -c "$sel" ${cfg}_min$mr < gpurun_out/r06h/${cfg}_min$mr.json || tail -3 gpurun_out/r06h/${cfg}_min$mr.err
  done
done
AB_STEPS=40 bash tools/gpu_ab.sh r06h "base:" "min32:GHN3_P8_MIN_ROWS=32" 2>&1 | tee gpurun_out/r06h/ab.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or resnet or vit" 2>&1 | tail -3
