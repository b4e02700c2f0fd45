timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
pick() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), 'fps ms/step', round(d['ms_per_step'],3), 'pyr_ms', round(d['stage_ms_per_step']['pyramid_ms'],3), 'frac', round(d['roofline']['frac'],3), 'match', round(d['last_match_ms'],3))"; }
$B 2>&1 | pick bench
$B --batch 128 2>&1 | pick bench128
