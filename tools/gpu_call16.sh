#!/bin/bash
# round 6, call 16: host-side trims of the fused layers (raw stream handle, integer pointers, dbeta | dgamma written in place)
set -u
mkdir -p gpurun_out/r06y
timeout 1500 python -m pytest tests/test_gpu_target_ops.py tests/test_gpu_networks.py tests/test_gpu_trainer.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20 | tee gpurun_out/r06y/tests_host_trims.txt
timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step" | sed "s/^/warm-up: /" | tee gpurun_out/r06y/train_host_trims.txt
for rep in 1 2 3; do
  timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step|Error|error" | tail -2 | sed "s/^/pass=$rep: /" | tee -a gpurun_out/r06y/train_host_trims.txt
done
GHN3_CPROFILE=gpurun_out/r06y/cprofile_host_trims.txt timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step" | sed "s/^/cprofile: /" | tee -a gpurun_out/r06y/train_host_trims.txt
