#!/bin/bash
# rocprofv3 kernel-trace summary of the bench workload (run on the GPU box through gpurun):
#   bash tools/profile_bench.sh <tag> [compute]      e.g.  bash tools/profile_bench.sh r01f f16
# Writes gpurun_out/<tag>_rocprof_kernel_stats_xl_<compute>.txt, <tag>_bench_under_rocprof_xl_<compute>.json and
# <tag>_kernel_timeline_last_step_xl_<compute>.csv (start / duration / queue of every kernel of the last step).
TAG=${1:-prof}; CT=${2:-f16}; EXTRA=${3:-}   # EXTRA: further bench.py flags, e.g. "--gpus 1 --force-ddp"
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS=5; WARM=2; NINST=5; NSER=5   # (bench.py: warm-up, timed steps, max(3, min(10, steps)) instrumented steps of the same schedule -- profile mode 2 --, then as many serialised, untimed steps: profile mode 3)
rm -rf /tmp/prof_$TAG
CMD="python3 bench.py --compute $CT --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extras $EXTRA"
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- $CMD > gpurun_out/${TAG}_bench_under_rocprof_xl_$CT.json 2> /tmp/prof_err.log
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
if [ -z "$DB" ]; then echo "no db"; tail -5 /tmp/prof_err.log; exit 1; fi
python3 tools/rocprof_summary.py "$DB" gpurun_out/${TAG}_rocprof_kernel_stats_xl_$CT.txt "rocprofv3 --kernel-trace --stats -- $CMD (ghn3xlm16, one 256-node graph, side stream on; the trace also holds $NINST instrumented steps of the same schedule and the $NSER serialised steps of the roofline pass)" $((STEPS+WARM+NINST+NSER)) $NSER
python3 tools/rocprof_timeline.py "$DB" gpurun_out/${TAG}_kernel_timeline_last_step_xl_$CT.csv 1 $((NSER+NINST))      # the last TIMED step
head -30 gpurun_out/${TAG}_rocprof_kernel_stats_xl_$CT.txt | cut -c1-60,110-200
tail -1 gpurun_out/${TAG}_bench_under_rocprof_xl_$CT.json | cut -c1-300
