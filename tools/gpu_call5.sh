#!/bin/bash
# round 6: the training loop with the dense-convolution op on / off (GHN3_NATIVE_CONV), alternating, after one find-db warm-up pass
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06j
export MIOPEN_FIND_MODE=3
timeout 600 python examples/train_ghn_ddp.py --steps 23 2>&1 | grep -E "ms per step" | sed "s/^/warm-up (cold find-db): /" | tee gpurun_out/r06j/train_ab.txt
for rep in 1 2; do
for nc in 0 1; do
  GHN3_NATIVE_CONV=$nc timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step|Error|error" | tail -2 | sed "s/^/native_conv=$nc pass=$rep: /" | tee -a gpurun_out/r06j/train_ab.txt
done
done
