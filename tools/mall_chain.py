#!/usr/bin/env python3
"""Does producer -> consumer plane reuse hit the Infinity Cache (256 MiB, memory side) when octave 0's blur chain runs in image
groups instead of whole-batch launches?

The octave-0 chain of the benchmark (128 frames, 1280x960 planes, five scale blurs with 5/7/9/11/13 taps, each reading the plane
the previous launch wrote) is run on the pyramid's own layout (6 consecutive planes per image) as
    for group in groups of G images:  blur 1 .. blur 5 on that group         (round-robin over NS streams)
and timed as a whole (all 128 images). G = 128 is what the library does today: every launch streams 629 MB + 629 MB, so nothing
a launch writes is still on the die when the next one reads it. Run ON the GPU box:  python tools/mall_chain.py
"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vulkansift_amd import api  # noqa: E402

L = api.lib()


class Plane(C.Structure):   # vksift_hip_Plane (include/vksift_hip.h)
    _fields_ = [("base", C.c_void_p), ("w", C.c_uint32), ("h", C.c_uint32), ("pitch", C.c_uint32), ("img_stride", C.c_uint64), ("fp16", C.c_uint32), ("reverse", C.c_uint32)]


L.vksift_hip_blur.argtypes = [Plane, Plane, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_void_p]
L.vksift_hip_blur.restype = C.c_int
B, H, W, NPL = int(os.environ.get("B", 128)), int(os.environ.get("H", 960)), int(os.environ.get("W", 1280)), 6
NTAPS = [5, 7, 9, 11, 13]
pyr = torch.rand(B, NPL, H, W, device="cuda")
IMG = NPL * H * W
taps = {nt: (C.c_float * 32)(*([1.0 / (2 * nt - 1)] * nt)) for nt in NTAPS}


def plane(first, layer, reverse=0):
    return Plane(pyr.data_ptr() + 4 * (first * IMG + layer * H * W), W, H, W, IMG, 0, reverse)


ALT = False


def chain(G, streams, layers=5):
    ns = len(streams)
    for gi, first in enumerate(range(0, B, G)):
        n = min(G, B - first)
        s = streams[gi % ns].cuda_stream if ns else None
        for k in range(layers):
            e = L.vksift_hip_blur(plane(first, k), plane(first, k + 1, (k & 1) if ALT else 0), taps[NTAPS[k]], NTAPS[k], n, s)
            assert e == 0, e


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


if __name__ == "__main__":
    pool = [torch.cuda.Stream() for _ in range(4)]
    base = None
    print(f"chain of 5 blurs over {B} x {W}x{H} planes ({B * H * W * 4 / 1e6:.0f} MB per plane batch); 8 B/texel/launch algorithmic")
    for G, ns in [(B, 1), (64, 1), (32, 1), (16, 1), (8, 1), (4, 1), (32, 2), (16, 2), (8, 2), (4, 2), (16, 4), (8, 4), (4, 4), (2, 4)]:
        if G > B:
            continue
        ms = timeit(lambda: chain(G, pool[:ns]))
        base = base or ms
        gbs = 5 * 8 * B * H * W / ms / 1e6
        print(f"G={G:4d} streams={ns}  {ms:7.3f} ms  {gbs:7.0f} GB/s algorithmic  x{base / ms:.3f}  working set/launch {2 * G * H * W * 4 / 1e6:.0f} MB, chain {6 * G * H * W * 4 / 1e6:.0f} MB")
    print("alternating dispatch direction (launch k+1 starts where launch k ended):")
    ALT = True
    for G, ns in [(B, 1), (64, 1), (32, 1), (16, 2), (64, 2), (32, 2)]:
        if G > B:
            continue
        ms = timeit(lambda: chain(G, pool[:ns]))
        print(f"ALT G={G:4d} streams={ns}  {ms:7.3f} ms  {5 * 8 * B * H * W / ms / 1e6:7.0f} GB/s algorithmic  x{base / ms:.3f}")
    ALT = False
    # one launch pair in isolation: consumer right behind its producer, by group size (the pure producer->consumer effect)
    for G in (128, 32, 16, 8, 4):
        if G > B:
            continue
        ms = timeit(lambda: chain(G, pool[:1], layers=2))
        print(f"two-launch chain G={G:4d}: {ms:7.3f} ms  {2 * 8 * B * H * W / ms / 1e6:7.0f} GB/s")


def warm_vs_cold():
    """The cache effect alone, launch geometry held fixed: time ONE consumer launch over G images whose source planes were (warm)
    written by the launch right before it, or (cold) written, then pushed out of every cache by a 2 GB fill. Also plain torch
    copies of the same bytes, as the ceiling for a 1-read-1-write stream out of the cache."""
    flush = torch.empty(512 << 20, dtype=torch.float32, device="cuda")     # 2 GB
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def one(G, nt, cold, torch_copy=False):
        ts = []
        for _ in range(7):
            if torch_copy:
                pyr[:G, 0].mul_(1.0)                                        # producer: writes plane 0 of G images
            else:
                assert L.vksift_hip_blur(plane(0, 2), plane(0, 0), taps[5], 5, G, None) == 0
            if cold:
                flush.fill_(1.0)
            ev0.record()
            if torch_copy:
                pyr[:G, 1].copy_(pyr[:G, 0])
            else:
                assert L.vksift_hip_blur(plane(0, 0), plane(0, 1), taps[nt], nt, G, None) == 0
            ev1.record()
            torch.cuda.synchronize()
            ts.append(ev0.elapsed_time(ev1) * 1e3)
        return sorted(ts)[len(ts) // 2]

    print("consumer launch alone: source planes warm (just written) vs cold (2 GB fill in between); us and GB/s at 8 B/texel")
    for G in (4, 8, 16, 24, 32, 64, 128):
        if G > B:
            continue
        mb = G * H * W * 4 / 1e6
        row = [f"G={G:4d} ({mb:5.0f} MB/plane batch)"]
        for name, nt, tc in (("blur5", 5, False), ("blur9", 9, False), ("blur13", 13, False), ("torch copy", 0, True)):
            w, c = one(G, nt, False, tc), one(G, nt, True, tc)
            row.append(f"{name}: warm {w:6.1f} us {8 * G * H * W / w / 1e3:5.0f} | cold {c:6.1f} us {8 * G * H * W / c / 1e3:5.0f}")
        print("  ".join(row))


if __name__ == "__main__" and os.environ.get("WARM_COLD"):
    warm_vs_cold()
