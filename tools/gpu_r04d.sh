cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
run() { name=$1; shift
  env "$@" python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r04d/$name.json 2>> gpurun_out/r04d/err.txt
  python - $name <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r04d/%s.json' % sys.argv[1]).read().strip().split('\n')[-1])
k = d['roofline']['kernels']
print('%-22s %.3f ms  frac %.3f  wgrad %.3f dgrad %.3f fwd %.3f tile_fwd %.3f tile_bwd %.3f' % (sys.argv[1], d['ms_per_step'], d['roofline']['frac'], k['w2_wgrad']['ms_per_step'], k['w2_dgrad']['ms_per_step'], k['w2_fwd']['ms_per_step'], k['tile_fwd']['ms_per_step'], k['tile_bwd']['ms_per_step']))
PY
}
run fused GHN3_X=1
run unfused GHN3_FUSED_LOSS=0
run fused_main GHN3_WGRAD_MAIN=1
run fused_cap160 GHN3_WGRAD_CAP=160
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_trainer.py -x -q 2>&1 | tail -12 > gpurun_out/r04d/tests.txt
cat gpurun_out/r04d/tests.txt; tail -5 gpurun_out/r04d/err.txt
