#!/usr/bin/env python3
"""Does the relative placement of the source / destination / DoG planes matter (HBM channel interleaving)? Times the
octave-0 blur of the benchmark shape with the three buffers carved out of one allocation at controlled relative offsets."""
import ctypes as C, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vulkansift_amd import api
L = api.lib()

class Plane(C.Structure):
    _fields_ = [("base", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32), ("pitch", C.c_uint32), ("img_stride", C.c_uint64)]

L.vksift_hip_blur.argtypes = [Plane, Plane, Plane, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_void_p]
B, H, W = 128, 960, 1280
n = B * H * W
big = torch.rand(3 * n + (1 << 24), device="cuda")
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
nt = 5
taps = (C.c_float * 32)(*([1.0 / (2 * nt - 1)] * nt))
for img_pad in (0, 64, 1088, 16448):
    for delta in (0, 64, 192, 1024 + 64, 16384 + 64, 4096):
        stride = H * W + img_pad
        if B * stride * 3 + 2 * delta > big.numel():
            continue
        base = big.data_ptr()
        def pl(k):
            return Plane(base + 4 * (k * (B * stride + delta)), W, H, W, stride)
        us = timeit(lambda: L.vksift_hip_blur(pl(0), pl(1), pl(2), taps, nt, B, None))
        print("img_pad(floats)", img_pad, "buffer delta(floats)", delta, "us", round(us, 1), "GB/s", round(12 * n / us / 1e3))
