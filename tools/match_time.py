"""Time the device-pointer matcher (vksift_hip_match_2nn_desc) on synthetic descriptors: usage match_time.py <rows> [repeats]"""
import sys, torch
sys.path.insert(0, '/root/repo')
from vulkansift_amd import api, multigpu
api.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 20
a = torch.from_numpy(api.gen_synthetic_descriptors(1, n)).cuda(); b = torch.from_numpy(api.gen_synthetic_descriptors(2, n)).cuda()
for _ in range(3):
    multigpu.hip_match_fn(a, 0, b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(rep):
    multigpu.hip_match_fn(a, 0, b)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / rep
print(f"{n} x {n}: {ms:.4f} ms  {2.0*n*n*128/ms/1e9:.0f} TOPS  {2.0*n*n*128/ms/1e9/3944*100:.1f} %")
