B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
sel='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); k=d["roofline"].get("kernels",{}); print(sys.argv[1], round(d["ms_per_step"],3), round(d["roofline"]["frac"],4), k["w2_wgrad"]["ms_per_step"])'
for cap in 224 256; do for tpw in 0 1 2 4; do GHN3_WGRAD_CAP=$cap GHN3_WGRAD_TPW=$tpw $B 2>/dev/null | python -c "$sel" cap${cap}_tpw$tpw; done; done
