timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
VKSIFT_BLUR_LEAN=0 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k pyramid 2>&1 | tail -2
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
pick() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), 'fps ms/step', round(d['ms_per_step'],3), 'pyr_ms', round(d['stage_ms_per_step']['pyramid_ms'],3), 'frac', round(d['roofline']['frac'],3), 'total', round(d['stage_ms_per_step']['total_ms'],3))"; }
$B 2>&1 | pick lazy
VKSIFT_LAZY_TOP=0 $B 2>&1 | pick nolazy
VKSIFT_COARSE_AFTER=1 $B 2>&1 | pick lazy_coarse_after
$B 2>&1 | pick lazy
VKSIFT_LAZY_TOP=0 $B 2>&1 | pick nolazy
VKSIFT_COARSE_AFTER=1 $B 2>&1 | pick lazy_coarse_after
