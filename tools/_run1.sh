cd $GRAFT_REPO_ROOT
timeout 900 python tests/gpu_accuracy.py 2>&1 | grep -v amdgpu | tail -10
