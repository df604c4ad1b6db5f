cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04i
run() { name=$1; shift
  env "$@" python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r04i/$name.json 2>> gpurun_out/r04i/err.txt
  python - $name <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r04i/%s.json' % sys.argv[1]).read().strip().split('\n')[-1])
k = d['roofline']['kernels']
print('%-26s %.3f ms  frac %.3f  wgrad %.3f' % (sys.argv[1], d['ms_per_step'], d['roofline']['frac'], k['w2_wgrad']['ms_per_step']))
PY
}
run base_cap128 GHN3_X=1
run base_cap0 GHN3_LAYER_WGRAD_CAP=0
run base_cap64 GHN3_LAYER_WGRAD_CAP=64
run main_cap128 GHN3_WGRAD_MAIN=1
run main_cap0 GHN3_WGRAD_MAIN=1 GHN3_LAYER_WGRAD_CAP=0
run main_cap64 GHN3_WGRAD_MAIN=1 GHN3_LAYER_WGRAD_CAP=64
run main_cap192 GHN3_WGRAD_MAIN=1 GHN3_LAYER_WGRAD_CAP=192
run w160_cap128 GHN3_WGRAD_CAP=160
tail -3 gpurun_out/r04i/err.txt
