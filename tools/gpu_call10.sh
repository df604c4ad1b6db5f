#!/bin/bash
# round 6, call 10: conv pair + stems on the dense op -- tests, then the training loop A/B on one box
set -u
mkdir -p gpurun_out/r06y
timeout 1500 python -m pytest tests/test_gpu_target_ops.py tests/test_gpu_networks.py tests/test_gpu_trainer.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20 | tee gpurun_out/r06y/tests_pair_stem.txt
timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step" | sed "s/^/warm-up: /" | tee gpurun_out/r06y/train_pair_stem_ab.txt
for rep in 1 2 3; do
for nat in 0 1; do
  GHN3_NATIVE_STEM=$nat GHN3_NATIVE_PAIR=$nat timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step|Error|error" | tail -2 | sed "s/^/pair+stem native=$nat pass=$rep: /" | tee -a gpurun_out/r06y/train_pair_stem_ab.txt
done
done
GHN3_CPROFILE=gpurun_out/r06y/cprofile_pair_stem.txt timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step" | sed "s/^/cprofile: /" | tee -a gpurun_out/r06y/train_pair_stem_ab.txt
