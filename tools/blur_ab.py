#!/usr/bin/env python3
"""A/B of the blur launch forms on octave 0's shape (B x 1280x960 fp32 planes, default B = 512 as in the bench), in ONE process with
the variants interleaved repetition by repetition (the boxes drift by several per cent over seconds): for every tap count the
two-texels-per-lane form (k_blur_lean) and the four-texels-per-lane form (k_blur_wide), each at several launch sizes, through the
development knobs of vksift_hip_tune(). Prints median / minimum time per launch, the rate at 8 B/texel, and checks that the output plane
does not depend on the variant. Run ON the GPU box:  python tools/blur_ab.py [taps ...]   (WG="5120,10240,20480" REPS=30 B=512)"""
import ctypes as C, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vulkansift_amd import api

L = api.lib()


class Plane(C.Structure):   # vksift_hip_Plane (include/vksift_hip.h)
    _fields_ = [("base", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32), ("pitch", C.c_uint32), ("img_stride", C.c_uint64), ("fp16", C.c_uint32),
                ("reverse", C.c_uint32)]


L.vksift_hip_blur.argtypes = [Plane, Plane, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_void_p]
L.vksift_hip_blur.restype = C.c_int
L.vksift_hip_tune.argtypes = [C.c_int, C.c_int]
B, H, W = int(os.environ.get("B", 512)), int(os.environ.get("H", 960)), int(os.environ.get("W", 1280))
REPS = int(os.environ.get("REPS", 24))
WGS = [int(x) for x in os.environ.get("WG", "10240").split(",")]
torch.manual_seed(1)
# production geometry: image i of the batch IS floats behind image i - 1 (the pyramid's image stride for 640x480 frames), the destination
# plane one plane behind the source inside the image; PAD_GB of memory allocated first, so that the planes land beyond the slow low range
# of the device's memory (the library's instances pick their range by measurement)
IS = int(os.environ.get("IS", 12938176))
PS = H * W
pad = torch.empty(int(float(os.environ.get("PAD_GB", 80)) * (1 << 30)), dtype=torch.uint8, device="cuda") if float(os.environ.get("PAD_GB", 80)) > 0 else None
arena = torch.empty(B * IS + 2 * PS, device="cuda")
imgs = arena[: B * IS].view(B, IS)
src = imgs[:, :PS].view(B, H, W)
dst = imgs[:, PS:2 * PS].view(B, H, W)
src.copy_(torch.rand(B, H, W, device="cuda"))


def plane(t, rev=0):
    return Plane(t.data_ptr(), W, H, W, IS, 0, rev)


def checksum(t):
    return int(t.contiguous().view(torch.int32).to(torch.int64).sum().item()) & 0xFFFFFFFFFFFF


for nt in [int(x) for x in (sys.argv[1:] or ["5", "7", "9", "11", "13"])]:
    w = [2.0 ** (-abs(i) / 2.0) for i in range(nt)]
    norm = w[0] + 2 * sum(w[1:])
    taps = (C.c_float * 32)(*[x / norm for x in w])
    variants = [(name, mask, wg) for wg in WGS for name, mask in (("lean", 0), ("wide", 0xFFFFF))]
    chk = {}
    for v in variants:
        L.vksift_hip_tune(1, v[1]); L.vksift_hip_tune(0, v[2])
        dst.zero_()
        assert L.vksift_hip_blur(plane(src), plane(dst), taps, nt, B, None) == 0
        torch.cuda.synchronize()
        chk[v] = checksum(dst)
    assert len(set(chk.values())) == 1, chk
    times = {v: [] for v in variants}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for rep in range(REPS):
        for v in variants:
            L.vksift_hip_tune(1, v[1]); L.vksift_hip_tune(0, v[2])
            ev[0].record()
            L.vksift_hip_blur(plane(src, rep & 1), plane(dst, rep & 1), taps, nt, B, None)
            ev[1].record()
            torch.cuda.synchronize()
            times[v].append(ev[0].elapsed_time(ev[1]) * 1e3)
    for v in variants:
        ts = sorted(times[v][2:])
        us = ts[len(ts) // 2]
        print("nt=%2d %-5s wg=%6d  %8.1f us (min %8.1f)  %5.0f GB/s  frac %.3f" % (nt, v[0], v[2], us, ts[0], 8 * B * H * W / us / 1e3, 8 * B * H * W / us / 8e6), flush=True)
    print("        identical planes: chk %012x" % list(chk.values())[0], flush=True)
L.vksift_hip_tune(1, -1); L.vksift_hip_tune(0, 0)

