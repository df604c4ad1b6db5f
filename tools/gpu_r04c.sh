cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r04c/$name.json 2>> gpurun_out/r04c/err.txt
  python - $name <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r04c/%s.json' % sys.argv[1]).read().strip().split('\n')[-1])
k = d['roofline']['kernels']
print('%-22s %.3f ms  frac %.3f  wgrad %.3f dgrad %.3f fwd %.3f' % (sys.argv[1], d['ms_per_step'], d['roofline']['frac'], k['w2_wgrad']['ms_per_step'], k['w2_dgrad']['ms_per_step'], k['w2_fwd']['ms_per_step']))
PY
}
run base GHN3_X=1
run cap96 GHN3_WGRAD_CAP=96
run cap128 GHN3_WGRAD_CAP=128
run cap160 GHN3_WGRAD_CAP=160
run cap192 GHN3_WGRAD_CAP=192
run cap256 GHN3_WGRAD_CAP=256
run wg_main GHN3_WGRAD_MAIN=1
run noside GHN3_NO_SIDE_STREAM=1
run old_base GHN3_X3S=0
run old_cap128 GHN3_X3S=0 GHN3_WGRAD_CAP=128
