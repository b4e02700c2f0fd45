timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['stage_ms_per_step']['descriptor_ms'])"; done
