#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for b in 100000000 2500000 1500000 700000; do
  export GHN3_XCD_B_BYTES=$b
  echo "== B budget $b"
  timeout 600 python tests/gemm_bench.py bf16 wgrad 2>&1 | grep "tile=25" | grep "K=   768"
  bash tools/pmc_profile.sh r02zb > /dev/null 2>&1; grep "h16w" gpurun_out/r02zb_pmc_xl_f16.txt | head -1
done
