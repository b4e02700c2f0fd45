timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pyramid or features" 2>&1 | tail -4
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
pick() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), 'fps pyr_ms', round(d['stage_ms_per_step']['pyramid_ms'],3), 'frac', round(d['roofline']['frac'],3))"; }
$B 2>&1 | pick lean
VKSIFT_BLUR_LEAN=0 $B 2>&1 | pick old
VKSIFT_BLUR_WGS=4096 $B 2>&1 | pick lean_wg4096
VKSIFT_BLUR_WGS=8192 VKSIFT_BLUR_MIN_SEG=32 $B 2>&1 | pick lean_wg8192_seg32
cd /tmp; export TMPDIR=/tmp
VKSIFT_SERIAL_OCTAVES=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/lean -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py gpurun_out/lean k_blur > gpurun_out/lean.txt
