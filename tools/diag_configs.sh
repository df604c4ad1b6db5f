#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
O=gpurun_out/r02zf_train_warm2.txt; : > $O
for m in 3 3 2; do
  echo "== MIOPEN_FIND_MODE=$m --amp --steps 63" >> $O
  MIOPEN_FIND_MODE=$m timeout 1200 python examples/train_ghn_ddp.py --amp --steps 63 2>&1 | tail -1 >> $O
done
echo "== MIOPEN_FIND_MODE=3 --steps 63 (fp32) twice" >> $O
MIOPEN_FIND_MODE=3 timeout 1200 python examples/train_ghn_ddp.py --steps 63 2>&1 | tail -1 >> $O
MIOPEN_FIND_MODE=3 timeout 1200 python examples/train_ghn_ddp.py --steps 63 2>&1 | tail -1 >> $O
cat $O
