B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
pick() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), 'fps ms/step', round(d['ms_per_step'],3), 'pyr_ms', round(d['stage_ms_per_step']['pyramid_ms'],3), 'frac', round(d['roofline']['frac'],3), 'total', round(d['stage_ms_per_step']['total_ms'],3))"; }
for i in 1 2; do
$B 2>&1 | pick prio
VKSIFT_STREAM_PRIO=0 $B 2>&1 | pick noprio
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prio -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py gpurun_out/prio > gpurun_out/prio_timeline.txt
