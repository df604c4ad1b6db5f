cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -s -k "vit or resnet" 2>&1 | grep -E "passed|failed|rror|worst|assert" | head -20
