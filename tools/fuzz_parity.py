#!/usr/bin/env python3
"""Extra randomised detection / matching parity against the oracle with fresh seeds (the committed tests use fixed ones).
usage (on the GPU box): python tools/fuzz_parity.py <seed> <cases> [max_side [min_side]]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vulkansift_amd import api as vk
from oracle import oracle
vk.lib().vksift_setLogLevel(vk.VKSIFT_LOG_ERROR)
seed, cases = int(sys.argv[1]), int(sys.argv[2])
max_side = int(sys.argv[3]) if len(sys.argv) > 3 else 1100
min_side = int(sys.argv[4]) if len(sys.argv) > 4 else 64
rng = np.random.default_rng(seed)
bad = 0
for case in range(cases):
    w, h = int(rng.integers(min_side, max_side)), int(rng.integers(min_side, max(min_side + 1, max_side * 3 // 4)))
    if w * h < 1024:      # below the API's minimum image size (vulkansift.c:600)
        h = 1024 // w + 1
    kw = {"seed_scale_sigma": float(np.float32(rng.uniform(1.2, 2.8))), "input_image_blur_level": float(np.float32(rng.uniform(0.3, 0.6))),
          "intensity_threshold": float(np.float32(rng.uniform(0.01, 0.08))), "edge_threshold": float(np.float32(rng.uniform(4.0, 16.0)))}
    if rng.random() < 0.5:
        kw["use_input_upsampling"] = False
    if rng.random() < 0.5:
        kw["nb_scales_per_octave"] = int(rng.integers(1, 9))
    if rng.random() < 0.3:
        kw["use_hardware_interpolated_blur"] = False
    if rng.random() < 0.3:
        kw["max_nb_orientation_per_keypoint"] = int(rng.integers(1, 5))
    if rng.random() < 0.3:
        kw["descriptor_format"] = vk.VKSIFT_DESCRIPTOR_FORMAT_VLFEAT
    if rng.random() < 0.2:
        kw["max_nb_sift_per_buffer"] = int(rng.integers(50, 2000))
    fp16 = rng.random() < 0.25                    # the defined FP16 pyramid mode against the oracle's model of it
    nb = int(rng.choice([1, 2, 3, 8, 9, 13]))
    if w * h * nb > 3_000_000:
        nb = 1
    okw, vkw = {}, {}
    for k, v in dict(kw, input_image_max_size=max(w * h, 128 * 128)).items():   # the oracle's config spells two fields differently
        if k in ("use_input_upsampling", "use_hardware_interpolated_blur"):
            okw[k], vkw[k] = int(v), bool(v)
        elif k == "descriptor_format":
            okw["use_vlfeat_format"], vkw[k] = int(v), int(v)
        else:
            okw[k] = vkw[k] = v
    if fp16:
        vkw["pyramid_precision_mode"], okw["pyramid_fp16"] = 1, 1
    vcfg = vk.default_config(sift_buffer_count=nb, **vkw)
    ocfg = oracle.default_config(math_mode=1, **okw)
    imgs = [vk.gen_synthetic_image(seed * 1000 + 17 * case + i, w, h) for i in range(nb)]
    with vk.Instance(vcfg, batch_capacity=nb) as inst:
        inst.detectFeaturesBatch(imgs, 0)
        feats = [inst.downloadFeatures(i) for i in range(nb)]
        if nb >= 2:
            inst.matchFeaturesBatch(list(range(nb)), [(i + 1) % nb for i in range(nb)])
            ms = [inst.downloadMatchesBatch(k) for k in range(nb)]
    refs = [oracle.detect(ocfg, im)[0] for im in imgs]
    ok = all(f.tobytes() == r.tobytes() for f, r in zip(feats, refs))
    if ok and nb >= 2 and min(len(r) for r in refs) >= 2:
        for k in range(nb):
            rm = oracle.match_2nn(refs[k], refs[(k + 1) % nb])
            ok = ok and len(ms[k]) == len(rm) and ms[k].tobytes() == rm.tobytes()
    if not ok:
        bad += 1
        print("MISMATCH", case, w, h, nb, kw, "fp16" if fp16 else "fp32", [len(f) for f in feats], [len(r) for r in refs])
print("cases", cases, "bad", bad, "features", sum(len(r) for r in refs))
sys.exit(1 if bad else 0)
