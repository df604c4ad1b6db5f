#!/bin/bash
# PMC counters of the target-network kernels (run on the GPU box through gpurun): one rocprofv3 pass per counter set over the
# micro-benchmark of the dense-convolution op (tools/tnet_conv_bench.py, REPS=5).
#   bash tools/tnet_pmc.sh  ->  gpurun_out/r06y/tnet_pmc.txt
set -u
mkdir -p gpurun_out/r06y
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06y/tnet_pmc.txt
echo "# rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/tnet_conv_bench.py (REPS=5; 12 shapes of the training loop, forward and forward + backward; sums over the run)" > $OUT
cd /tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/tpmc_$name
  REPS=5 timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/tpmc_$name -o r -- python3 $ROOT/tools/tnet_conv_bench.py > /tmp/tpmc.log 2> /tmp/tpmc_err.log
  DB=$(find /tmp/tpmc_$name -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    python3 $ROOT/tools/rocprof_pmc_summary.py "$DB" "$(echo $set | cut -c1-20)" tnet_ | head -12 >> $OUT
  else
    echo "no db for $set" >> $OUT; tail -3 /tmp/tpmc_err.log >> $OUT
  fi
done
cat $OUT | cut -c1-330
