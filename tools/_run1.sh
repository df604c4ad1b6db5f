cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "0 0" "16 0" "24 0" "0 16" "0 24"; do
set -- $cfg
GHN3_FWD_TILE=$1 GHN3_DGRAD_TILE=$2 timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/b_x.log 2>&1
python - <<PY
import json
for l in open('gpurun_out/b_x.log'):
    if l.startswith('{'):
        d = json.loads(l)
        k = d['roofline']['kernels']
        print("$cfg", d['ms_per_step'], k['w2_fwd']['ms_per_step'], k['w2_dgrad']['ms_per_step'])
PY
done
