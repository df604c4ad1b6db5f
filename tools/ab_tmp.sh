B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
sel='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); k=d["roofline"].get("kernels",{}); print(sys.argv[1], round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), {a:k[a]["ms_per_step"] for a in ("w2_fwd","w2_dgrad","w2_wgrad")})'
$B 2>/dev/null | python -c "$sel" variant
bash tools/pmc_profile.sh r03r_var > /dev/null 2>&1
python - <<EOF2
import json
d=json.load(open("gpurun_out/r03r_var_pmc_traffic_xl_f16.json"))
print("traffic", round(d["hbm_bytes_per_step"]/1e9,3), {k[:24]:(round(v["fetch"]*2048/1e9,2),round(v["write"]*1024/1e9,2)) for k,v in d["per_kernel_raw_kib_per_step"].items()})
EOF2
