B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
sel='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); k=d["roofline"].get("kernels",{}); print(sys.argv[1], round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), k["w2_wgrad"]["ms_per_step"])'
for bb in 1572864 2500000 5000000 800000; do GHN3_XCD_B_BYTES=$bb $B 2>/dev/null | python -c "$sel" bbytes$bb; done
