#!/bin/bash
# round 6, call 13: second-version dense convolution kernels (GHN3_TNET_CONV2) -- whole-network tests, training loop A/B, and the
# kernel trace of the loop
set -u
mkdir -p gpurun_out/r06y
timeout 1500 python -m pytest tests/test_gpu_networks.py tests/test_gpu_trainer.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20 | tee gpurun_out/r06y/tests_conv2.txt
timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step" | sed "s/^/warm-up: /" | tee gpurun_out/r06y/train_conv2_ab.txt
for rep in 1 2 3; do
for v in 0 1; do
  GHN3_TNET_CONV2=$v timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step|Error|error" | tail -2 | sed "s/^/conv2=$v pass=$rep: /" | tee -a gpurun_out/r06y/train_conv2_ab.txt
done
done
bash tools/gpu_call11.sh 2>&1 | tail -42
