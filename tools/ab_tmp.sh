sel='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], "ms", round(d["ms_per_step"],3), "G/s", round(d["value"]/1e9,2), "frac", round(d["roofline"]["frac"],4), "tf", round(d["roofline"]["achieved"],1))'
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
$B --model ghn3xlm16 --nodes 200 2>/dev/null | python -c "$sel" xl_200
$B --model ghn3lm8 --nodes 200 2>/dev/null | python -c "$sel" lm8_200
$B --model ghn3lm8 --nodes 200 --graphs-per-gpu 8 2>/dev/null | python -c "$sel" lm8_200_x8
$B --model ghn3tm8 --nodes 128 2>/dev/null | python -c "$sel" tm8_128_f16
$B --model ghn3tm8 --nodes 128 --compute bf16 2>/dev/null | python -c "$sel" tm8_128_bf16
$B --model ghn3sm8 --nodes 128 2>/dev/null | python -c "$sel" sm8_128
