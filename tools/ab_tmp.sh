B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
sel='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], round(d["ms_per_step"],3), round(d["roofline"]["frac"],4))'
for i in 1 2 3; do $B 2>/dev/null | python -c "$sel" cast_fast; done
