cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "split_bf16 or gemm_cases" 2>&1 | tail -15
timeout 900 python tests/x3_bench.py 2>&1 | grep -v amdgpu.ids
