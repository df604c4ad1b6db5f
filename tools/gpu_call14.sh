#!/bin/bash
# round 6, call 14: second versions of the weight-gradient kernels of the target-network ops -- tests, micro-benchmark, loop
set -u
mkdir -p gpurun_out/r06y
timeout 1500 python -m pytest tests/test_gpu_target_ops.py tests/test_gpu_networks.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20 | tee gpurun_out/r06y/tests_wgrad2.txt
python tools/tnet_conv_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06y/tnet_conv_bench_v3.txt
timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step" | sed "s/^/warm-up: /" | tee gpurun_out/r06y/train_wgrad2.txt
for rep in 1 2 3; do
  timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step|Error|error" | tail -2 | sed "s/^/pass=$rep: /" | tee -a gpurun_out/r06y/train_wgrad2.txt
done
bash tools/gpu_call11.sh 2>&1 | grep -E "tnet_conv_wgrad|tnet_dw_wgrad|tnet_pw_wgrad|GPU busy"
