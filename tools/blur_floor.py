#!/usr/bin/env python3
"""Time vksift_hip_blur alone (octave-0 shape of the benchmark: 128 x 1280x960 planes, 1 read + 1 write per texel) for several
tap counts, against plain torch kernels on the same buffers (copy, scaled copy): how far is the blur from what the memory
system gives a 1-read-1-write stream? Run ON the GPU box."""
import ctypes as C, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vulkansift_amd import api
L = api.lib()


class Plane(C.Structure):   # vksift_hip_Plane (include/vksift_hip.h)
    _fields_ = [("base", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32), ("pitch", C.c_uint32), ("img_stride", C.c_uint64), ("fp16", C.c_uint32)]


L.vksift_hip_blur.argtypes = [Plane, Plane, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_void_p]
L.vksift_hip_blur.restype = C.c_int
B, H, W = int(os.environ.get("B", 128)), 960, 1280
src = torch.rand(B, H, W, device="cuda"); dst = torch.empty_like(src)
def plane(t): return Plane(t.data_ptr(), W, H, W, H * W, 0)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
for nt in (2, 3, 5, 7, 9, 11, 13):
    taps = (C.c_float * 32)(*([1.0 / (2 * nt - 1)] * nt))
    assert L.vksift_hip_blur(plane(src), plane(dst), taps, nt, B, None) == 0
    us = timeit(lambda: L.vksift_hip_blur(plane(src), plane(dst), taps, nt, B, None))
    print("blur nt=%d" % nt, round(us, 1), "us", round(8 * B * H * W / us / 1e3), "GB/s (8 B/texel)")
us = timeit(lambda: torch.mul(src, 2.0, out=dst))
print("torch mul (8 B/texel)", round(us, 1), "us", round(8 * B * H * W / us / 1e3), "GB/s")
us = timeit(lambda: dst.copy_(src))
print("torch copy (8 B/texel)", round(us, 1), "us", round(8 * B * H * W / us / 1e3), "GB/s")
