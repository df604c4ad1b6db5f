#!/bin/bash
# round 6, call 1: new-code sanity (tile heights 7 / 9, detach fix, f16 saturation, bias gather) + A/B of the balanced row tiles
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06a
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cast16_and_16bit or split_bf16 or f16_piece or tiny_forward" 2>&1 | tail -5
  timeout 1200 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "overlapped or against_the_reference or properties_at_full or xlm16_f16_forward" 2>&1 | tail -5 ) > gpurun_out/r06a/tests.log 2>&1
tail -12 gpurun_out/r06a/tests.log
AB_STEPS=40 bash tools/gpu_ab.sh r06a base:GHN3_P8_BALANCED=0 bal:GHN3_P8_BALANCED=1 2>&1 | tee gpurun_out/r06a/ab.log
