#!/bin/bash
# A/B lines on one GPU box: bash tools/gpu_ab.sh <tag> "<name>:<ENV=.. ENV=..>" ...   (name "base" = no variables)
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/$TAG
B="python bench.py --steps ${AB_STEPS:-60} --warmup 5 --no-cpu-baseline --no-extras ${AB_FLAGS:-}"   # AB_FLAGS: further bench.py flags
sel='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); k=d["roofline"]["kernels"]; ki=d["roofline"].get("kernels_in_step", k); print(sys.argv[1], "ms", round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],4), "frac_step", round(d["roofline"].get("frac_step", 0),4), "cap", d["roofline"].get("wgrad_side_workgroups"), " ".join("%s %.3f" % (n, k[n]["ms_per_step"]) for n in ("w2_fwd","w2_dgrad","w2_wgrad","tile_fwd","tile_bwd")), "in-step wgrad %.3f" % ki["w2_wgrad"]["ms_per_step"])'
for rep in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  [ "$envs" = "$spec" ] && envs=""
  env $envs $B > gpurun_out/$TAG/${name}_$rep.json 2> gpurun_out/$TAG/${name}_$rep.err
  python -c "$sel" ${name}_$rep < gpurun_out/$TAG/${name}_$rep.json || tail -3 gpurun_out/$TAG/${name}_$rep.err
done
done
