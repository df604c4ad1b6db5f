sel='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], json.dumps(d["cpu_baseline"]))'
for th in 256 64 16; do GHN3_CPU_THREADS=$th python bench.py --steps 3 --warmup 2 --no-extras 2>/dev/null | python -c "$sel" threads$th; done
