#!/bin/bash
# round 6, call 12: two operand pieces instead of three in the target-network kernels (GHN3_TNET_TERMS) -- tests at the same
# tolerances, then the training loop A/B on one box
set -u
mkdir -p gpurun_out/r06y
GHN3_TNET_TERMS=2 timeout 1500 python -m pytest tests/test_gpu_target_ops.py tests/test_gpu_networks.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20 | tee gpurun_out/r06y/tests_terms2.txt
timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step" | sed "s/^/warm-up: /" | tee gpurun_out/r06y/train_terms_ab.txt
for rep in 1 2 3; do
for tm in 3 2; do
  GHN3_TNET_TERMS=$tm timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step|Error|error" | tail -2 | sed "s/^/terms=$tm pass=$rep: /" | tee -a gpurun_out/r06y/train_terms_ab.txt
done
done
