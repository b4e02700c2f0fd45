timeout 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
VKSIFT_SERIAL_OCTAVES=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/span -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $GRAFT_REPO_ROOT/gpurun_out/span k_xx | sed -n 2,4p
cd $GRAFT_REPO_ROOT
for i in 1 2; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120; done
