cd "${GRAFT_REPO_ROOT:-.}"
for args in "ghn3lm8 f16 25 40" "ghn3xlm16 f16 40"; do
  timeout 600 python tests/gpu_diag_configs.py $args 2>&1 | grep -v amdgpu.ids | head -8 | cut -c1-160
done
