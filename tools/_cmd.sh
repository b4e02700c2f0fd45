timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
pick() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), 'fps ms/step', round(d['ms_per_step'],3), 'pyr_ms', round(d['stage_ms_per_step']['pyramid_ms'],3), 'frac', round(d['roofline']['frac'],3), 'total', round(d['stage_ms_per_step']['total_ms'],3))"; }
$B 2>&1 | pick new
$B 2>&1 | pick new
cd /tmp; export TMPDIR=/tmp
VKSIFT_SERIAL_OCTAVES=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ext2 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py gpurun_out/ext2 k_extrema | tail -6
