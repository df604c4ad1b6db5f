B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
sel='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); k=d["roofline"].get("kernels",{}); print(sys.argv[1], round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), {a:k[a]["ms_per_step"] for a in ("fc_bwd","w0_bwd","w2_wgrad")})'
GHN3_FC_DGRAD_T=0 $B 2>/dev/null | python -c "$sel" fcT0
$B 2>/dev/null | python -c "$sel" fcT1
GHN3_FC_DGRAD_T=0 $B 2>/dev/null | python -c "$sel" fcT0
$B 2>/dev/null | python -c "$sel" fcT1
