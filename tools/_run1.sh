cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -s -k "resnet" 2>&1 | tail -25 > gpurun_out/t16.log
cat gpurun_out/t16.log
