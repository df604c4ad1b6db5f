#!/bin/bash
# round 6, call 7: the GHN host compile in the loader workers (GraphBatch.precompile) -- training loop A/B on one box,
# the new GPU test, and a cProfile of the loop with the hand-off on
set -u
mkdir -p gpurun_out/r06w
timeout 900 python -m pytest tests/test_gpu_trainer.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r06w/tests.txt
timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step" | sed "s/^/warm-up: /" | tee gpurun_out/r06w/train_ab.txt
for rep in 1 2 3; do
for wc in 0 1; do
  GHN3_WORKER_COMPILE=$wc timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step|Error|error" | tail -2 | sed "s/^/worker_compile=$wc pass=$rep: /" | tee -a gpurun_out/r06w/train_ab.txt
done
done
GHN3_CPROFILE=gpurun_out/r06w/cprofile_worker_compile.txt timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step" | sed "s/^/cprofile: /" | tee -a gpurun_out/r06w/train_ab.txt
