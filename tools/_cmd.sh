B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
pick() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']), 'fps ms/step', round(d['ms_per_step'],3), 'pyr_ms', round(d['stage_ms_per_step']['pyramid_ms'],3), 'frac', round(d['roofline']['frac'],3), 'total', round(d['stage_ms_per_step']['total_ms'],3))"; }
for i in 1 2; do
VKSIFT_LIB=$PWD/vulkansift_amd/lib/libvulkansift_base.so $B 2>&1 | pick base
VKSIFT_LIB=$PWD/vulkansift_amd/lib/libvulkansift_ilp.so $B 2>&1 | pick ilp
done
VKSIFT_LIB=$PWD/vulkansift_amd/lib/libvulkansift_ilp.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k pyramid 2>&1 | tail -2
