python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "match" 2>&1 | tail -2
for f in 0 1; do VKSIFT_MATCH_FORM=$f python - <<'PY'
import sys, torch, os
sys.path.insert(0,'.')
from vulkansift_amd import api, multigpu
api.lib().vksift_setLogLevel(4)
api.load()
for n in (50000, 100000, 13000, 2000):
    a=torch.from_numpy(api.gen_synthetic_descriptors(1,n)).cuda(); b=torch.from_numpy(api.gen_synthetic_descriptors(2,n)).cuda()
    multigpu.hip_match_fn(a,0,b); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): multigpu.hip_match_fn(a,0,b)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(os.environ.get("VKSIFT_MATCH_FORM"), n, "ms", round(ms,4), "TOPS", round(2*n*n*128/ms/1e9,1))
PY
done
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), round(d['roofline']['frac'],3), d['last_match_ms'])"
