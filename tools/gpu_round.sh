#!/bin/bash
# One GPU-box call that produces the judged artefacts of a round (run through gpurun):
#   bash tools/gpu_round.sh <tag>
# -> gpurun_out/<tag>_bench_default_xl_f16.json (the default `python bench.py` line, incl. roofline / cpu_baseline / extras),
#    <tag>_rocprof_kernel_stats_xl_f16.txt + <tag>_kernel_timeline_last_step_xl_f16.csv (tools/profile_bench.sh),
#    <tag>_pmc_xl_f16.txt + <tag>_pmc_traffic_xl_f16.json (tools/pmc_profile.sh)
TAG=${1:-r}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
bash tools/profile_bench.sh $TAG f16 > gpurun_out/${TAG}_profile.log 2>&1
# the same profile with the side stream serialised into the chain: the kernels' OWN durations, what bench.py's `roofline`
# measures in its serialised pass (profile mode 3) -- the two must agree
GHN3_NO_SIDE_STREAM=1 bash tools/profile_bench.sh ${TAG}_serialised f16 > gpurun_out/${TAG}_profile_serialised.log 2>&1
bash tools/pmc_profile.sh $TAG > gpurun_out/${TAG}_pmc.log 2>&1
cp gpurun_out/${TAG}_pmc_traffic_xl_f16.json profiles/ 2>/dev/null     # (bench.py reads the newest committed traffic figure)
python bench.py > gpurun_out/${TAG}_bench_default_xl_f16.json 2> gpurun_out/${TAG}_bench_default.err
tail -c 2500 gpurun_out/${TAG}_bench_default_xl_f16.json
