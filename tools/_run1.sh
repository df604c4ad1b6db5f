cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1
tail -5 gpurun_out/bench_default.log | cut -c1-2500
