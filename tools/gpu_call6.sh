#!/bin/bash
# round 6: lazy layout (fused layers keep NHWC, stock light layers convert) on / off: network parity tests + the training loop
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06v
timeout 900 python -m pytest tests/test_gpu_target_ops.py tests/test_gpu_networks.py tests/test_gpu_trainer.py -x -q -m gpu 2>&1 | tail -6
export MIOPEN_FIND_MODE=3
timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step" | sed "s/^/warm-up: /" | tee gpurun_out/r06v/train_ab.txt
for rep in 1 2; do
for lz in 0 1; do
  GHN3_NATIVE_LAZY_LAYOUT=$lz timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step|Error|error" | tail -2 | sed "s/^/lazy_layout=$lz pass=$rep: /" | tee -a gpurun_out/r06v/train_ab.txt
done
done
