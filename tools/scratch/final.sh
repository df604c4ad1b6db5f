python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/profile_bench.sh r01l f16 > /dev/null 2>&1
bash tools/pmc_profile.sh r01l > /dev/null 2>&1
cp gpurun_out/r01l_pmc_traffic_xl_f16.json profiles/
python bench.py > gpurun_out/r01l_bench_default_xl_f16.json 2>/dev/null
tail -c 1800 gpurun_out/r01l_bench_default_xl_f16.json
python bench.py --graphs-per-gpu 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r01l_bench_xl_f16_b2.json
python bench.py --compute f32 --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > gpurun_out/r01l_bench_xl_f32.json
