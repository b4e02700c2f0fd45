#!/usr/bin/env python3
"""vksift_hip_blur_multi against vksift_hip_blur on the planes of ONE 640x480 detection (1280x960 ... 160x120), batch 1: time per launch
sequence (HIP events, median of REPS), 11 and 13 taps. Run on the GPU box: python tools/multi_blur_probe.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vulkansift_amd import api
L = api.lib()
class Plane(C.Structure):
    _fields_ = [("base", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32), ("pitch", C.c_uint32), ("img_stride", C.c_uint64), ("fp16", C.c_uint32), ("reverse", C.c_uint32)]
L.vksift_hip_blur.argtypes = [Plane, Plane, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_void_p]
L.vksift_hip_blur_multi.argtypes = [C.POINTER(Plane), C.POINTER(Plane), C.c_uint32, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_void_p]
L.vksift_hip_blur_multi.restype = C.c_int
B = int(os.environ.get("B", 1)); REPS = int(os.environ.get("REPS", 200))
shapes = [(1280, 960), (640, 480), (320, 240), (160, 120)]
src = [torch.rand(B, h, w, device="cuda") for w, h in shapes]
dst = [torch.empty_like(s) for s in src]
def planes(ts): return [Plane(t.data_ptr(), t.shape[2], t.shape[1], t.shape[2], t.shape[1] * t.shape[2], 0, 0) for t in ts]
ps, pd = planes(src), planes(dst)
def time(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for r in range(REPS + 20):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if r >= 20: ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2], ts[0]
for nt in (11, 13):
    w = [2.0 ** (-abs(i) / 3.0) for i in range(nt)]; n = w[0] + 2 * sum(w[1:]); t = (C.c_float * 32)(*[x / n for x in w])
    for lo, hi in ((0, 1), (0, 2), (0, 4), (1, 2), (1, 4)):
        k = hi - lo
        A, D = (Plane * k)(*ps[lo:hi]), (Plane * k)(*pd[lo:hi])
        one = time(lambda: [L.vksift_hip_blur(ps[i], pd[i], t, nt, B, None) for i in range(lo, hi)])
        ref = [d.clone() for d in dst[lo:hi]]
        mul = time(lambda: L.vksift_hip_blur_multi(A, D, k, t, nt, B, None))
        same = all(torch.equal(a, b) for a, b in zip(ref, dst[lo:hi]))
        print(f"taps {nt} octaves {lo}..{hi - 1}: per-octave launches {one[0]:.1f} us (min {one[1]:.1f}), one multi launch {mul[0]:.1f} us (min {mul[1]:.1f}), identical {same}")
