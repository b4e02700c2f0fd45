"""Time the device-pointer matcher (vksift_hip_match_2nn_desc) on synthetic descriptors: usage match_time.py <rows> [repeats]"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkansift_amd import api, multigpu
api.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 20
a = torch.from_numpy(api.gen_synthetic_descriptors(1, n)).cuda(); b = torch.from_numpy(api.gen_synthetic_descriptors(2, n)).cuda()
import os, zlib
L = api.lib(); L.vksift_hip_tune.argtypes = [__import__("ctypes").c_int] * 2
for kv in os.environ.get("TUNE", "").split(","):
    if kv:
        L.vksift_hip_tune(int(kv.split("=")[0]), int(kv.split("=")[1]))
for _ in range(3):
    rec = multigpu.hip_match_fn(a, 0, b)
print("crc", zlib.crc32(rec.cpu().numpy().tobytes()) & 0xFFFFFFFF)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(rep):
    multigpu.hip_match_fn(a, 0, b)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / rep
print(f"{n} x {n}: {ms:.4f} ms  {2.0*n*n*128/ms/1e9:.0f} TOPS  {2.0*n*n*128/ms/1e9/3944*100:.1f} %")
