cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | head
