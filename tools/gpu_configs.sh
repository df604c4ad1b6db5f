#!/bin/bash
# Bench lines + rocprofv3 kernel summaries of BASELINE.json configs 1-4 (run on the GPU box through gpurun):
#   bash tools/gpu_configs.sh <tag>
# -> gpurun_out/<tag>_bench_<config>.json (python bench.py --config <config>: the line incl. roofline + cpu_baseline) and
#    gpurun_out/<tag>_rocprof_kernel_stats_<config>.txt (rocprofv3 --kernel-trace --stats of the same command, short)
TAG=${1:-cfg}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for CFG in resnet18-tm8 resnet50-xl vit-xl tm8-128 lm8-200; do
  python bench.py --config $CFG --steps 50 --warmup 5 > gpurun_out/${TAG}_bench_${CFG}.json 2> gpurun_out/${TAG}_bench_${CFG}.err
  tail -c 600 gpurun_out/${TAG}_bench_${CFG}.json | head -c 400; echo
  rm -rf /tmp/prof_$CFG
  STEPS=10; WARM=3
  CMD="python3 bench.py --config $CFG --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extras"
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$CFG -o r -- $CMD > /tmp/prof_${CFG}.json 2> /tmp/prof_${CFG}.err
  DB=$(find /tmp/prof_$CFG -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    python3 tools/rocprof_summary.py "$DB" gpurun_out/${TAG}_rocprof_kernel_stats_${CFG}.txt "rocprofv3 --kernel-trace --stats -- $CMD" $((STEPS+WARM))
  else
    echo "no db for $CFG"; tail -3 /tmp/prof_${CFG}.err
  fi
done
