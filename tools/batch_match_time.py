"""Time the batched self-match of B 640x480 frames alone (no detection running beside it): usage batch_match_time.py [B] [repeats]"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from vulkansift_amd import api
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W, H = 640, 480
api.lib().vksift_setLogLevel(api.VKSIFT_LOG_ERROR)
gen = np.stack([api.gen_synthetic_image(0x5EED0000 + i, W, H) for i in range(min(B, 64))])
host = np.ascontiguousarray(np.concatenate([gen] * ((B + len(gen) - 1) // len(gen)))[:B])
d = torch.from_numpy(host).cuda()
cfg = api.default_config(sift_buffer_count=B, gpu_device_index=0, input_image_max_size=W * H)
inst = api.Instance(cfg, batch_capacity=B)
inst.detectFeaturesBatchDevice(d.data_ptr(), B, W, H, 0)
ids = list(range(B))
inst.matchFeaturesBatch(ids, ids)
torch.cuda.synchronize()
nf = np.mean([inst.getFeaturesNumber(i) for i in range(0, B, 16)])
t0 = time.perf_counter()
for _ in range(rep):
    inst.matchFeaturesBatch(ids, ids)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / rep
ops = 2.0 * B * nf * nf * 128
print(f"batch match {B} x ({nf:.0f} x {nf:.0f}): {dt*1e3:.3f} ms per call  {ops/dt/1e12:.0f} TOPS  {ops/dt/1e12/3944*100:.1f} %")
if len(sys.argv) > 3 and sys.argv[3] == "match-only":
    inst.close()
    sys.exit(0)
t0 = time.perf_counter()
for _ in range(rep):
    inst.detectFeaturesBatchDevice(d.data_ptr(), B, W, H, 0)
torch.cuda.synchronize()
print(f"detect only: {(time.perf_counter()-t0)/rep*1e3:.3f} ms per call")
t0 = time.perf_counter()
for _ in range(rep):
    inst.detectFeaturesBatchDevice(d.data_ptr(), B, W, H, 0)
    inst.matchFeaturesBatch(ids, ids)
torch.cuda.synchronize()
print(f"detect + match: {(time.perf_counter()-t0)/rep*1e3:.3f} ms per call")
inst.close()
