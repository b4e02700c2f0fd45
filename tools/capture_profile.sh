#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel trace + the two PMC passes first, the report (profiles/<tag>_*) from them, then the
# official bench line — in that order so that the line's roofline.traffic comes from the PMC capture of the same kernel sources.
# usage: bash tools/capture_profile.sh <tag> [width height batch]   -> gpurun_out/<tag>/{trace,pmc_fetch,pmc_write,profiles/,bench.json}
TAG=${1:-r02}; W=${2:-640}; H=${3:-480}; B=${4:-128}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -- python $R/bench.py --no-cpu-baseline --no-extras --steps 10 > $OUT/trace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1
cd $R
python tools/make_profile_report.py gpurun_out/$TAG $TAG $W $H $B > $OUT/report.log 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cp $OUT/bench.json profiles/${TAG}_bench.json
mkdir -p $OUT/profiles; cp profiles/${TAG}_* $OUT/profiles/
# the raw traces are large: only the report travels back
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
tail -c 1500 $OUT/report.log; tail -1 $OUT/bench.json | cut -c 1-1500
