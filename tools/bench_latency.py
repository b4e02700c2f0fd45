#!/usr/bin/env python3
"""Single-image latency, the reference's own published measurement (src/perf/perf_runtime.cpp:63-81 through
src/perf/wrappers/vulkansift_wrapper.cpp:30-33): mean wall-clock of
    vksift_detectFeatures + vksift_getFeaturesNumber + vksift_downloadFeatures
for one host image, default config, 10 warm-up + 100 measured runs (docs/Performances.md:22,43).
Reference figures (other hardware): 640x480 ~4.5-7 ms, 1536x1024 10.6-16.5 ms, 3456x2304 46-73 ms.
usage: python tools/bench_latency.py [WxH ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(w, h, warm=int(os.environ.get("WARM", 10)), runs=int(os.environ.get("RUNS", 100))):
    from vulkansift_amd import api
    api.lib().vksift_setLogLevel(api.VKSIFT_LOG_ERROR)
    for kv in filter(None, os.environ.get("TUNE", "").split(",")):   # development: "knob=value,..." for vksift_hip_tune
        api.lib().vksift_hip_tune(int(kv.split("=")[0]), int(kv.split("=")[1]))
    img = api.gen_synthetic_image(0xABC0 + w, w, h)
    cfg = api.default_config(input_image_max_size=w * h)
    with api.Instance(cfg) as inst:
        n = 0
        for i in range(warm + runs):
            if i == warm:
                t0 = time.perf_counter()
            inst.detectFeatures(img, 0)
            n = inst.getFeaturesNumber(0)
            inst.downloadFeatures(0)
        dt = (time.perf_counter() - t0) / runs
    return {"image": f"{w}x{h}", "features": int(n), "latency_ms": dt * 1e3, "frames_per_s": 1.0 / dt}


def run_match(w, h, warm=10, runs=100):
    """vksift_matchFeatures + vksift_getMatchesNumber + vksift_downloadMatches of two detected images (the reference's
    src/examples/test_sift_match.cpp flow; it publishes no matching time)."""
    from vulkansift_amd import api
    img1, img2 = api.gen_synthetic_image(0xABD0 + w, w, h), api.gen_synthetic_image(0xABE0 + w, w, h)
    cfg = api.default_config(input_image_max_size=w * h)
    with api.Instance(cfg) as inst:
        inst.detectFeatures(img1, 0)
        inst.detectFeatures(img2, 1)
        n1, n2 = inst.getFeaturesNumber(0), inst.getFeaturesNumber(1)
        for i in range(warm + runs):
            if i == warm:
                t0 = time.perf_counter()
            inst.matchFeatures(0, 1)
            inst.downloadMatches()
        dt = (time.perf_counter() - t0) / runs
    return {"match": f"{n1}x{n2} features ({w}x{h} images)", "latency_ms": dt * 1e3}


if __name__ == "__main__":
    sizes = [a for a in sys.argv[1:] if "x" in a] or ["640x480", "1536x1024", "3456x2304"]
    for sz in sizes:
        w, h = map(int, sz.split("x"))
        print(json.dumps(run(w, h)), flush=True)
    for sz in sizes:
        w, h = map(int, sz.split("x"))
        print(json.dumps(run_match(w, h)), flush=True)
