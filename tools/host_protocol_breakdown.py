#!/usr/bin/env python3
"""Where the time of the reference's measurement protocol goes (host image in, features + matches out), 128 x 640x480."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vulkansift_amd import api
import torch

api.lib().vksift_setLogLevel(api.VKSIFT_LOG_ERROR)
B, W, H = 128, 640, 480
frames = [api.gen_synthetic_image(0x5EED0000 + i, W, H) for i in range(B)]
cfg = api.default_config(sift_buffer_count=B, input_image_max_size=W * H)
inst = api.Instance(cfg, batch_capacity=B)
lib = api.lib()
feat = np.zeros(cfg.max_nb_sift_per_buffer, api.FEATURE_DTYPE)
mt = np.zeros(cfg.max_nb_sift_per_buffer, api.MATCH_DTYPE)
def T(f, n=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def det(): inst.detectFeaturesBatch(frames, 0)
def det_sync():
    inst.detectFeaturesBatch(frames, 0); lib.vksift_getFeaturesNumber(inst._h, 0)
def dl():
    for i in range(B):
        lib.vksift_getFeaturesNumber(inst._h, i); lib.vksift_downloadFeatures(inst._h, feat.ctypes.data, i)
def match():
    for i0 in range(0, B, 64):
        ids = list(range(i0, i0 + 64)); inst.matchFeaturesBatch(ids, ids)
        for k in range(64):
            lib.vksift_ext_getMatchesNumberBatch(inst._h, k); lib.vksift_ext_downloadMatchesBatch(inst._h, k, mt.ctypes.data)
t0 = time.perf_counter(); inst.detectFeaturesBatch(frames, 0); t_enq = (time.perf_counter() - t0) * 1e3
print("first detectFeaturesBatch(host) returns after %.2f ms" % t_enq)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); inst.detectFeaturesBatch(frames, 0); ts.append((time.perf_counter() - t0) * 1e3); torch.cuda.synchronize()
print("detectFeaturesBatch(host) returns after %.2f ms (staging + enqueue, GPU idle before); usable cores %d" % (min(ts), len(os.sched_getaffinity(0))))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cgroup cpu.max unavailable", e)
ptrs = (__import__("ctypes").c_void_p * B)(*[f.ctypes.data for f in frames])
ts = []
for _ in range(5):
    t0 = time.perf_counter(); lib.vksift_ext_detectFeaturesBatch(inst._h, ptrs, B, W, H, 0); ts.append((time.perf_counter() - t0) * 1e3); torch.cuda.synchronize()
print("the C call alone (no Python wrapper work): %.2f ms" % min(ts))
print("detect (host images) + wait: %.2f ms" % T(det_sync))
print("128 x (count + downloadFeatures): %.2f ms" % T(dl))
print("2 x matchFeaturesBatch(64) + 128 x downloadMatches: %.2f ms" % T(match))
inst.close()

# ---- the pipelined protocol (bench.py: value_host_input_pipelined), host-side phase times per batch
cfg2 = api.default_config(sift_buffer_count=2 * B, input_image_max_size=W * H)
inst = api.Instance(cfg2, batch_capacity=B)
ids = [list(range(B)), list(range(B, 2 * B))]
ph = {"detect_call": [], "feats": [], "matches": [], "match_call": []}
def tick(name, t0):
    t1 = time.perf_counter(); ph[name].append((t1 - t0) * 1e3); return t1
n = 8
inst.detectFeaturesBatch(frames, 0); inst.matchFeaturesBatch(ids[0], ids[0])
t_start = time.perf_counter()
for it in range(n):
    cur, nxt = it & 1, (it & 1) ^ 1
    t = time.perf_counter()
    if it + 1 < n:
        inst.detectFeaturesBatch(frames, nxt * B)
    t = tick("detect_call", t)
    for i in ids[cur]:
        lib.vksift_getFeaturesNumber(inst._h, i); lib.vksift_downloadFeatures(inst._h, feat.ctypes.data, i)
    t = tick("feats", t)
    for k in range(B):
        lib.vksift_ext_getMatchesNumberBatch(inst._h, k); lib.vksift_ext_downloadMatchesBatch(inst._h, k, mt.ctypes.data)
    t = tick("matches", t)
    if it + 1 < n:
        inst.matchFeaturesBatch(ids[nxt], ids[nxt])
    t = tick("match_call", t)
total = (time.perf_counter() - t_start) * 1e3
print("pipelined: %.2f ms per batch of %d; host phases (median ms): %s" % (total / n, B, {k: round(sorted(v)[len(v) // 2], 2) for k, v in ph.items()}))
inst.close()
