#!/usr/bin/env python3
"""The four metrics of the reference's src/perf/perf_matching.cpp on five synthetic warps, HIP path and CPU oracle side by side.
Run on the GPU box: python tools/quality_report.py [width height]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import quality
from oracle import oracle as O
from vulkansift_amd import api as vk

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
vk.lib().vksift_setLogLevel(vk.VKSIFT_LOG_ERROR)
img1 = vk.gen_synthetic_image(35, w, h)
ocfg = O.default_config(math_mode=0)      # independent libm math on the CPU side
o1, _ = O.detect(ocfg, img1)
print("| warp | path | kp1 | kp2 | matches | repeatability | putative match ratio | precision | matching score |")
print("|---|---|---|---|---|---|---|---|---|")
with vk.Instance(vk.default_config(input_image_max_size=max(w * h, 1024))) as inst:
    for k, kw in enumerate(quality.WARPS):
        H = quality.homography(w, h, **kw)
        img2 = quality.warp(img1, H)
        inst.detectFeatures(img1, 0); inst.detectFeatures(img2, 1)
        inst.matchFeaturesFiltered([0], [1], 0.75, True)
        fm = inst.downloadFilteredMatches(0)
        f1, f2 = inst.downloadFeatures(0), inst.downloadFeatures(1)
        s1 = quality.score(f1, f2, fm["idx_a"], fm["idx_b"], H, w, h)
        o2, _ = O.detect(ocfg, img2)
        ia, ib = O.filter_matches(O.match_2nn(o1, o2), O.match_2nn(o2, o1), 0.75, True)
        s2 = quality.score(o1, o2, ia, ib, H, w, h)
        for name, s in (("HIP", s1), ("oracle (libm)", s2)):
            print(f"| {k + 1}: rot {kw['angle_deg']:.0f} deg, x{kw['scale']} | {name} | {s['keypoints_1']} | {s['keypoints_2']} | {s['matches']} | {s['repeatability']:.3f} | "
                  f"{s['putative_match_ratio']:.3f} | {s['precision']:.3f} | {s['matching_score']:.3f} |")
