#!/bin/bash
# average duration of the seed launch from a kernel trace of bench.py
for v in "" vulkansift_amd/lib/variants/lib_pyr_lut.so "" vulkansift_amd/lib/variants/lib_pyr_lut.so; do
  cd /tmp; export TMPDIR=/tmp
  VKSIFT_LIB=${v:+$GRAFT_REPO_ROOT/$v} rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tr -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 10 > /tmp/b.json 2>/dev/null
  cd $GRAFT_REPO_ROOT
  echo "== ${v:-new}: $(python -c "import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print(round(d['value']), d['stage_ms_per_call']['pyramid_ms'], d['stage_ms_per_call']['pyramid_all_ms'])")"
  python tools/prof_summary.py gpurun_out/tr "k_blur_lean<5, 1" 2>/dev/null | grep "k_blur_lean<5, 1" | head -1 | cut -c 60-140
  rm -rf gpurun_out/tr
done
