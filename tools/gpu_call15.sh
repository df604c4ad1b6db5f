#!/bin/bash
# round 6, call 15: hardware bf16 conversion in every target-network kernel -- tests, loop timing, kernel trace
set -u
mkdir -p gpurun_out/r06y
timeout 1500 python -m pytest tests/test_gpu_target_ops.py tests/test_gpu_networks.py tests/test_gpu_trainer.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20 | tee gpurun_out/r06y/tests_hwcvt.txt
python tools/tnet_conv_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06y/tnet_conv_bench_v4.txt | tail -3
timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step" | sed "s/^/warm-up: /" | tee gpurun_out/r06y/train_hwcvt.txt
for rep in 1 2 3; do
  timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step|Error|error" | tail -2 | sed "s/^/pass=$rep: /" | tee -a gpurun_out/r06y/train_hwcvt.txt
done
bash tools/gpu_call11.sh 2>&1 | head -36 | cut -c1-100,113-170
