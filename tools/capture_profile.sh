#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel trace + the two PMC passes first, the report (profiles/<tag>_*) from them, then the
# official bench line — in that order so that the line's roofline.traffic comes from the PMC capture of the same kernel sources.
# usage: bash tools/capture_profile.sh <tag> [width height batch]   -> gpurun_out/<tag>/{trace,pmc_fetch,pmc_write,profiles/,bench.json}
TAG=${1:-r04}; W=${2:-640}; H=${3:-480}; B=${4:-512}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -- python $R/bench.py --no-cpu-baseline --no-extras --steps 10 > $OUT/trace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1
cd $R
python tools/make_profile_report.py gpurun_out/$TAG $TAG $W $H $B > $OUT/report.log 2>&1
# BASELINE config 3 (64 x 1080p): traffic of its blur + scan launches, same two counters
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/c3/pmc_fetch -- python $R/tools/c3_leg.py 1 > $OUT/c3_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/c3/pmc_write -- python $R/tools/c3_leg.py 1 > $OUT/c3_write.log 2>&1
cd $R
python tools/make_profile_report.py gpurun_out/$TAG/c3 ${TAG}_c3 1920 1080 64 > $OUT/report_c3.log 2>&1
# the binary16 scale-space mode: traffic of its launches, same two counters (bench.py --fp16 reads <tag>_fp16_pmc_traffic.json)
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/fp16/pmc_fetch -- python $R/bench.py --fp16 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/fp16_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/fp16/pmc_write -- python $R/bench.py --fp16 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/fp16_write.log 2>&1
cd $R
python tools/make_profile_report.py gpurun_out/$TAG/fp16 ${TAG}_fp16 $W $H $B fp16 > $OUT/report_fp16.log 2>&1
rm -rf $OUT/fp16/pmc_fetch $OUT/fp16/pmc_write
# SQ counters of the VALU-bound kernels and of the seed / scan launches, detections one after the other (clean attribution)
VKSIFT_PYR_PINGPONG=0 PMC_GROUPS="SQ_WAVES,SQ_BUSY_CU_CYCLES,SQ_WAVE_CYCLES,SQ_INSTS_VALU;SQ_ACTIVE_INST_VALU,SQ_WAIT_INST_ANY,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE;SQ_VMEM_TA_ADDR_FIFO_FULL,SQ_INSTS_VMEM,SQ_INSTS_LDS,SQ_INSTS_SALU" PMC_PASS_TIMEOUT=240 \
  python tools/pmc_kernel.py "k_descriptor,k_orientation<,k_extrema_lean,k_blur_lean<5, 1,k_blur_pair_wide,k_blur_wide<13>@$(( ((2 * W + 127) / 128) * 64 ))" -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras \
  > profiles/${TAG}_sq_counters.json 2> $OUT/sq.log
timeout 1200 python tools/capture_match_counters.py $TAG > $OUT/match_counters.log 2>&1
# timeline of ONE single-image detection (launch count, durations, idle gaps): profiles/<tag>_latency_timeline.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/lat -- python $R/tools/bench_latency.py 640x480 > $OUT/lat.log 2>&1
cd $R
python tools/latency_timeline.py gpurun_out/$TAG/lat > profiles/${TAG}_latency_timeline.txt 2>&1
rm -rf $OUT/lat
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cp $OUT/bench.json profiles/${TAG}_bench.json
mkdir -p $OUT/profiles; cp profiles/${TAG}_* $OUT/profiles/
# the raw traces are large: only the report travels back
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/c3/pmc_fetch $OUT/c3/pmc_write
tail -c 1500 $OUT/report.log; tail -1 $OUT/bench.json | cut -c 1-1500
