pick() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), 'fps pyr_ms', round(d['stage_ms_per_step']['pyramid_ms'],3), 'frac', round(d['roofline']['frac'],3))"; }
for i in 1 2 3; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | pick default
VKSIFT_COARSE_AFTER=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | pick coarse_after
done
