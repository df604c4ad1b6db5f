#!/bin/bash
# The training loop of examples/train_ghn_ddp.py on one GPU box (run through gpurun): four passes from a cold box, one under
# cProfile, the same loop on the stock ATen / MIOpen layers (first and second pass), and the kernel trace (tools/train_loop_trace.sh).
#   bash tools/train_loop_final.sh   ->  gpurun_out/r06y/train_final.txt, cprofile_final.txt, train_loop_rocprof_kernel_stats.txt
set -u
mkdir -p gpurun_out/r06y
for rep in 1 2 3 4; do
  timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step|Error|error" | tail -1 | sed "s/^/pass=$rep: /" | tee -a gpurun_out/r06y/train_final.txt
done
GHN3_CPROFILE=gpurun_out/r06y/cprofile_final.txt timeout 600 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step" | sed "s/^/cprofile: /" | tee -a gpurun_out/r06y/train_final.txt
GHN3_NATIVE_OPS=0 timeout 900 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step" | sed "s/^/stock layers (GHN3_NATIVE_OPS=0), first pass: /" | tee -a gpurun_out/r06y/train_final.txt
GHN3_NATIVE_OPS=0 timeout 900 python examples/train_ghn_ddp.py --steps 63 2>&1 | grep -E "ms per step" | sed "s/^/stock layers (GHN3_NATIVE_OPS=0), second pass: /" | tee -a gpurun_out/r06y/train_final.txt
bash tools/train_loop_trace.sh 2>&1 | grep -E "GPU busy|kernel us/step"
