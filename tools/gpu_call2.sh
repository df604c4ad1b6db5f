#!/bin/bash
# round 6, call 2: FETCH_SIZE / L2 hit counters of the W2 forward / dgrad, balanced row tiles on and off
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06b
export TMPDIR=/tmp
export GHN3_NO_SIDE_STREAM=1
for bal in 0 1; do
  export GHN3_P8_BALANCED=$bal
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    name=$(echo $set | cut -d' ' -f1)
    rm -rf /tmp/pmc_$name
    timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$name -o r -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /tmp/pmc_bench.json 2> /tmp/pmc_err.log
    DB=$(find /tmp/pmc_$name -name "*.db" | head -1)
    if [ -n "$DB" ]; then python3 tools/pmc_p8_dispatches.py "$DB" gemm_p8_kernel 2 "balanced=$bal" >> gpurun_out/r06b/p8_counters.txt
    else echo "no db for $set (balanced=$bal)" >> gpurun_out/r06b/p8_counters.txt; tail -3 /tmp/pmc_err.log >> gpurun_out/r06b/p8_counters.txt; fi
  done
done
cat gpurun_out/r06b/p8_counters.txt
