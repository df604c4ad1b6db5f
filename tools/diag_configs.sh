#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
O=gpurun_out/r02s_train_scan.txt; : > $O
for a in "--meta-batch-size 8 --batch-size 64" "--meta-batch-size 8 --batch-size 8" "--meta-batch-size 1 --batch-size 64" "--meta-batch-size 8 --batch-size 64 --miopen --steps 63" "--meta-batch-size 8 --batch-size 64 --amp"; do
  echo "== $a" >> $O
  timeout 900 python examples/train_ghn_ddp.py --steps 23 $a 2>&1 | tail -1 >> $O
done
cat $O
