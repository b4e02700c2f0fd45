#!/usr/bin/env python3
"""Measurements of the other BASELINE.json configs on one GPU (not the driver contract; see bench.py for that).

  C3  batch of 64 1920x1080 frames, detect only -> pyramid+DoG algorithmic GB/s
  C4  2-NN brute force 50k x 50k descriptors (MFMA int8 formulation) -> time, int8 TOPS
  C5  one GPU's share of 512 x 1080p detect + pair matching (64 frames, 32 pairs)
usage: python tools/bench_configs.py [c3] [c4] [c5] [--batch N]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def c3(batch=64, ups=True, steps=5):
    import torch
    from vulkansift_amd import api

    api.lib().vksift_setLogLevel(api.VKSIFT_LOG_ERROR)
    W, H = 1920, 1080
    base = [api.gen_synthetic_image(0x5EED1000 + i, W, H) for i in range(8)]
    frames = np.stack([base[i % 8] for i in range(batch)])
    d = torch.from_numpy(frames).cuda()
    cfg = api.default_config(sift_buffer_count=batch, input_image_max_size=W * H, use_input_upsampling=ups)
    inst = api.Instance(cfg, batch_capacity=batch)
    inst.detectFeaturesBatchDevice(d.data_ptr(), batch, W, H, 0)
    torch.cuda.synchronize()
    nfeat = [inst.getFeaturesNumber(i) for i in range(min(batch, 8))]
    inst.setProfiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        inst.detectFeaturesBatchDevice(d.data_ptr(), batch, W, H, 0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    acc = inst.getAccumulatedDetectTimings()
    inst.close()
    pyr_s = acc["pyramid_ms"] * 1e-3
    out = {"config": "C3", "frames": batch, "upsampling": ups, "frames_per_s": batch * steps / dt,
           "pyramid_ms_per_batch": acc["pyramid_ms"] / acc["nb_calls"],
           "pyramid_algorithmic_GBps": acc["pyramid_algorithmic_bytes"] / pyr_s / 1e9,
           "frac_of_8TBps": acc["pyramid_algorithmic_bytes"] / pyr_s / 8e12,
           "stage_ms": {k: acc[k] / acc["nb_calls"] for k in ("pyramid_ms", "extrema_ms", "orientation_ms", "descriptor_ms", "total_ms")},
           "features_per_frame": float(np.mean(nfeat))}
    print(json.dumps(out))


def c5_share(batch=64, steps=4):
    """One GPU's share of BASELINE config C5 (512 x 1080p over 8 GPUs = 64 frames per GPU): detect (up-sampling ON) + 2-NN
    of the 32 consecutive frame pairs. The 8-GPU aggregate is the driver's bench run (weak scaling, no collectives)."""
    import torch
    from vulkansift_amd import api

    api.lib().vksift_setLogLevel(api.VKSIFT_LOG_ERROR)
    W, H = 1920, 1080
    base = [api.gen_synthetic_image(0x5EED1000 + i, W, H) for i in range(8)]
    frames = np.stack([base[i % 8] for i in range(batch)])
    d = torch.from_numpy(frames).cuda()
    cfg = api.default_config(sift_buffer_count=batch, input_image_max_size=W * H)
    inst = api.Instance(cfg, batch_capacity=batch)
    pa, pb = list(range(0, batch, 2)), list(range(1, batch, 2))

    def step():
        inst.detectFeaturesBatchDevice(d.data_ptr(), batch, W, H, 0)
        inst.matchFeaturesBatch(pa, pb)

    step()
    torch.cuda.synchronize()
    nfeat = [inst.getFeaturesNumber(i) for i in range(8)]
    nmatch = inst.getMatchesNumberBatch(0)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    inst.close()
    print(json.dumps({"config": "C5 (one GPU's share)", "frames": batch, "pairs": len(pa), "frames_per_s": batch * steps / dt,
                      "ms_per_batch": dt / steps * 1e3, "features_per_frame": float(np.mean(nfeat)), "matches_pair0": int(nmatch)}))


def c4(n=50000, steps=5):
    import torch
    from vulkansift_amd import api, multigpu

    a = torch.from_numpy(api.gen_synthetic_descriptors(1, n)).cuda()
    b = torch.from_numpy(api.gen_synthetic_descriptors(2, n)).cuda()
    multigpu.hip_match_fn(a, 0, b)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(steps):
        out = multigpu.hip_match_fn(a, 0, b)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / steps
    ops = 2.0 * n * n * 128
    print(json.dumps({"config": "C4", "n": n, "ms": ms, "int8_TOPS": ops / (ms * 1e-3) / 1e12, "matches_per_s": n / (ms * 1e-3),
                      "frac_of_3944_TOPS": ops / (ms * 1e-3) / 3944e12}))


if __name__ == "__main__":
    args = sys.argv[1:]
    batch = 64
    if "--batch" in args:
        batch = int(args[args.index("--batch") + 1])
    if not args or "c4" in args:
        c4()
    if not args or "c3" in args:
        c3(batch=batch)
        c3(batch=batch, ups=False)
    if not args or "c5" in args:
        c5_share(batch=batch)
