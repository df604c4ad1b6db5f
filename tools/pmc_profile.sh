#!/bin/bash
# PMC passes over the bench workload (run on the GPU box through gpurun):
#   bash tools/pmc_profile.sh <tag>          e.g.  bash tools/pmc_profile.sh r01f
# One rocprofv3 run per counter set (--kernel-trace + --pmc only, as the pool requires).  The side stream is
# serialised (GHN3_NO_SIDE_STREAM=1) so that the counters of concurrent kernels do not mix.  Writes
#   gpurun_out/<tag>_pmc_xl_f16.txt           per-kernel sums of every counter set
#   gpurun_out/<tag>_pmc_traffic_xl_f16.json  HBM bytes per step of the W2 GEMM family (bench.py roofline.traffic)
TAG=${1:-pmc}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
export GHN3_NO_SIDE_STREAM=1
OUT=gpurun_out/${TAG}_pmc_xl_f16.txt
STEPS=2; WARM=1; NINST=3; NSER=3       # (bench.py: warm-up + timed steps, then max(3, min(10, steps)) instrumented and as many serialised, untimed steps: they all run the same kernels)
: > $OUT
echo "# GHN3_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extras (ghn3xlm16 N=256, f16 mode; side stream serialised so that counters are per kernel); sums over the $((STEPS+WARM+NINST+NSER)) steps of the run (warm-up + timed + instrumented + the serialised roofline pass)" >> $OUT
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$name -o r -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extras > /tmp/pmc_bench.json 2> /tmp/pmc_err.log
  cp /tmp/pmc_bench.json /tmp/pmc_bench_$name.json
  DB=$(find /tmp/pmc_$name -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    python3 tools/rocprof_pmc_summary.py "$DB" "$(echo $set | cut -c1-20)" | head -14 >> $OUT
  else
    echo "no db for $set" >> $OUT; tail -3 /tmp/pmc_err.log >> $OUT
  fi
done
F=$(find /tmp/pmc_FETCH_SIZE -name "*.db" | head -1); W=$(find /tmp/pmc_WRITE_SIZE -name "*.db" | head -1)
python3 tools/pmc_traffic.py "$F" "$W" $((STEPS+WARM+NINST+NSER)) /tmp/pmc_bench_WRITE_SIZE.json > gpurun_out/${TAG}_pmc_traffic_xl_f16.json
cat gpurun_out/${TAG}_pmc_traffic_xl_f16.json
cut -c1-250 $OUT | head -40
