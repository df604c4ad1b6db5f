cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof16
rocprofv3 --kernel-trace --stats -d /tmp/prof16 -o r -- python3 bench.py --compute f16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench16.json 2> gpurun_out/prof_err.log
DB=$(find /tmp/prof16 -name "*.db" | head -1)
echo "db: $DB"
python3 tools/rocprof_summary.py "$DB" gpurun_out/prof16_stats.txt "rocprofv3 --kernel-trace --stats -- python3 bench.py --compute f16 --steps 5 --warmup 2 --no-cpu-baseline (ghn3xlm16, 256-node graph)" 7
head -45 gpurun_out/prof16_stats.txt | cut -c1-175
tail -2 gpurun_out/prof_bench16.json | cut -c1-300
