cd $GRAFT_REPO_ROOT
timeout 600 python examples/train_synthetic.py --steps 12 2>&1 | grep -v amdgpu | tail -8
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | head
