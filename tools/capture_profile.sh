#!/bin/bash
# Run on the GPU box (gpurun): official bench line + rocprofv3 kernel trace + the two PMC passes of the same command.
# usage: bash tools/capture_profile.sh <tag>     -> gpurun_out/<tag>/{bench.json,trace,pmc_fetch,pmc_write}
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python $R/bench.py --no-cpu-baseline --no-extras --steps 10 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1
cd $R
tail -1 $OUT/bench.json
