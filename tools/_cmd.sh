timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
pick() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), 'fps ms/step', round(d['ms_per_step'],3), 'pyr_ms', round(d['stage_ms_per_step']['pyramid_ms'],3), 'frac', round(d['roofline']['frac'],3), d['stage_ms_per_step'])"; }
$B 2>&1 | pick pipelined
VKSIFT_STAGE_SYNC=1 $B 2>&1 | pick stagesync
VKSIFT_BLUR_WGS=8192 VKSIFT_BLUR_MIN_SEG=32 $B 2>&1 | pick pipelined_wg8192_seg32
VKSIFT_BLUR_WGS=4096 VKSIFT_BLUR_MIN_SEG=32 $B 2>&1 | pick pipelined_wg4096_seg32
VKSIFT_BLUR_WGS=4096 VKSIFT_BLUR_MIN_SEG=64 $B 2>&1 | pick pipelined_wg4096_seg64
