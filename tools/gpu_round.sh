#!/bin/bash
# One GPU-box session: the GPU parity suite, the default bench line and the rocprof summary (run through gpurun):
#   bash tools/gpu_round.sh <tag> [tests|bench|prof|pmc ...]
TAG=${1:-r02}; shift
WHAT=${@:-tests bench prof}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for w in $WHAT; do
  case $w in
    tests)  timeout 1500 python -m pytest tests -m gpu -q -x --timeout 1200 2>&1 | tail -25 > gpurun_out/${TAG}_pytest.log; tail -8 gpurun_out/${TAG}_pytest.log ;;
    newtests) timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -q -s --timeout 1200 2>&1 | tail -60 > gpurun_out/${TAG}_pytest_configs.log; tail -40 gpurun_out/${TAG}_pytest_configs.log ;;
    bench)  timeout 900 python bench.py > gpurun_out/${TAG}_bench_default_xl_f16.json 2> gpurun_out/${TAG}_bench.err; tail -c 3000 gpurun_out/${TAG}_bench_default_xl_f16.json; tail -3 gpurun_out/${TAG}_bench.err ;;
    benchq) timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_bench_quick.json 2> gpurun_out/${TAG}_bench.err; tail -c 2500 gpurun_out/${TAG}_bench_quick.json; tail -3 gpurun_out/${TAG}_bench.err ;;
    prof)   bash tools/profile_bench.sh $TAG f16 ;;
    pmc)    bash tools/pmc_profile.sh $TAG ;;
  esac
done
