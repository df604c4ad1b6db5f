B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
sel='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], round(d["ms_per_step"],3), round(d["roofline"]["frac"],4))'
$B 2>/dev/null | python -c "$sel" baseline
GHN3_ABLATE_OPS=6 $B 2>/dev/null | python -c "$sel" no_ln_fwd
GHN3_ABLATE_OPS=14 $B 2>/dev/null | python -c "$sel" no_ln_bwd
GHN3_ABLATE_OPS=6,14 $B 2>/dev/null | python -c "$sel" no_ln_fwd_bwd
GHN3_ABLATE_OPS=15 $B 2>/dev/null | python -c "$sel" no_ln_param_grad
GHN3_ABLATE_OPS=7,16 $B 2>/dev/null | python -c "$sel" no_attention
$B 2>/dev/null | python -c "$sel" baseline
