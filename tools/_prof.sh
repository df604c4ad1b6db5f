cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in f16 f32; do
rm -rf /tmp/prof_$c
rocprofv3 --kernel-trace --stats -d /tmp/prof_$c -o r -- python3 bench.py --compute $c --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r01e_bench_under_rocprof_xl_$c.json 2> gpurun_out/prof_err_$c.log
DB=$(find /tmp/prof_$c -name "*.db" | head -1)
python3 tools/rocprof_summary.py "$DB" gpurun_out/r01e_rocprof_kernel_stats_xl_$c.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --compute $c --steps 5 --warmup 2 --no-cpu-baseline (ghn3xlm16, one 256-node graph, side stream on)" 7
done
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1
timeout 300 python bench.py --compute f32 --no-cpu-baseline > gpurun_out/bench_f32.log 2>&1
grep "^{" gpurun_out/bench_default.log | cut -c1-330
grep "^{" gpurun_out/bench_f32.log | cut -c1-300
head -16 gpurun_out/r01e_rocprof_kernel_stats_xl_f16.txt | cut -c1-170
