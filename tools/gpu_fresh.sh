#!/bin/bash
# fresh-architecture-per-step throughput under a few host settings (GPU box): bash tools/gpu_fresh.sh <tag>
TAG=${1:-fresh}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/$TAG
sel='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], "fixed", round(d["ms_per_step"],3), "fresh", round(d["fresh_graph_ms_per_step"],3), "gpu", round(d["fresh_graph_gpu_ms"],3), d["fresh_graph_host_ms"], "train", round(d["train_step"]["ms_per_step"],3))'
for spec in "w6:GHN3_LOADER_WORKERS=6" "w10:GHN3_LOADER_WORKERS=10" "w12:GHN3_LOADER_WORKERS=12" "w10thr:GHN3_LOADER_WORKERS=10 GHN3_FRESH_PREFETCH_THREAD=1"; do
  name=${spec%%:*}; envs=${spec#*:}
  env $envs python bench.py --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err
  python -c "$sel" $name < gpurun_out/$TAG/$name.json || tail -3 gpurun_out/$TAG/$name.err
done
