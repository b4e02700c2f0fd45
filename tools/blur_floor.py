#!/usr/bin/env python3
"""Time vksift_hip_blur alone (octave-0 shape of the benchmark: 128 x 1280x960 planes) for several tap counts, against a
plain 1-read-2-write torch kernel on the same buffers: how far is the blur from what the memory system gives this pattern?
Run ON the GPU box."""
import ctypes as C, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vulkansift_amd import api
L = api.lib()

class Plane(C.Structure):
    _fields_ = [("base", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32), ("pitch", C.c_uint32), ("img_stride", C.c_uint64)]

L.vksift_hip_blur.argtypes = [Plane, Plane, Plane, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_void_p]
L.vksift_hip_blur.restype = C.c_int
B, H, W = int(os.environ.get("B", 128)), 960, 1280
src = torch.rand(B, H, W, device="cuda"); dst = torch.empty_like(src); dog = torch.empty_like(src)
def plane(t): return Plane(t.data_ptr(), W, H, W, H * W)
none = Plane(0, 0, 0, 0, 0)
out = {}
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
for nt in (2, 3, 5, 7, 9, 11, 13):
    taps = (C.c_float * 32)(*([1.0 / (2 * nt - 1)] * nt))
    us = timeit(lambda: L.vksift_hip_blur(plane(src), plane(dst), plane(dog), taps, nt, B, None))
    out["blur+dog nt=%d" % nt] = {"us": round(us, 1), "alg_GBps": round(12 * B * H * W / us / 1e3, 1)}
    us = timeit(lambda: L.vksift_hip_blur(plane(src), plane(dst), none, taps, nt, B, None))
    out["blur only nt=%d" % nt] = {"us": round(us, 1), "alg_GBps": round(8 * B * H * W / us / 1e3, 1)}
us = timeit(lambda: (torch.mul(src, 2.0, out=dst), torch.sub(dst, src, out=dog)))
out["torch mul + sub (2 launches, 20 B/px)"] = {"us": round(us, 1), "GBps": round(20 * B * H * W / us / 1e3, 1)}
us = timeit(lambda: torch.mul(src, 2.0, out=dst))
out["torch mul (8 B/px)"] = {"us": round(us, 1), "GBps": round(8 * B * H * W / us / 1e3, 1)}
for k, v in out.items():
    print(k, v)
