cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "adamw" 2>&1 | grep -E "passed|failed|rror|assert" | head
