cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
export GHN3_NO_SIDE_STREAM=1
: > gpurun_out/r01e_pmc_xl_f16.txt
echo "# GHN3_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (ghn3xlm16 N=256, f16 mode; side stream serialised so that counters are per kernel); sums over the 3 steps of the run" >> gpurun_out/r01e_pmc_xl_f16.txt
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
rm -rf /tmp/pmc
timeout 600 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc -o r -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmc_bench.json 2> /tmp/pmc_err.log
DB=$(find /tmp/pmc -name "*.db" | head -1)
if [ -n "$DB" ]; then python3 tools/rocprof_pmc_summary.py "$DB" "$(echo $set | cut -c1-20)" | head -14 >> gpurun_out/r01e_pmc_xl_f16.txt; else echo "no db for $set"; tail -3 /tmp/pmc_err.log; fi
done
cat gpurun_out/r01e_pmc_xl_f16.txt | cut -c1-330
