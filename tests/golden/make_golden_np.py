#!/usr/bin/env python3
"""Second golden fixture set, produced WITHOUT the C oracle: the numpy restatements only
(tests/np_restatement.py: pyramid; tests/np_features.py: K4/K5/K6), from the image up.

    python tests/golden/make_golden_np.py
    python tests/golden/make_golden_np.py --bench-frame      (feats_c2_frame0_np.npy: frame 0 of the benchmark workload, 11 s)

Writes img_192x144.npy and feats_192x144_<config>_np.npy. The oracle (libm math) and the HIP path are both
compared against these files with the tolerances of tests/test_np_golden.py; a misreading of a shader would have
to be made twice, in two differently structured programs, to pass.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import np_features as NF      # noqa: E402
import np_restatement as R    # noqa: E402

CONFIGS = {
    "default": dict(),
    "noups_vlfeat": dict(ups=False, vlfeat=True, max_ori=0, interpolated=False),
    "s2": dict(S=2),
}


def image(seed=0x192, w=192, h=144):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.full((h, w), 128.0)
    for _ in range(int(w * h / 60)):
        cx, cy, s = rng.uniform(0, w), rng.uniform(0, h), np.exp(rng.uniform(np.log(0.8), np.log(8)))
        a = rng.uniform(25, 100) * rng.choice([-1, 1])
        img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    img += rng.uniform(-4, 4, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def detect(img, S=3, ups=True, interpolated=True, vlfeat=False, max_ori=4):
    pyr = R.build_pyramid(img, S=S, ups=ups, interpolated=interpolated)
    secs = [NF.detect_octave(g, d, S, o - (1 if ups else 0), max_nb_orientation=max_ori, vlfeat=vlfeat) for o, (g, d) in enumerate(pyr)]
    return np.concatenate(secs), [len(s) for s in secs]


def bench_frame():
    """frame 0 of bench.py's workload (BASELINE config 2: 640x480, default configuration), from the numpy restatements alone. The
    image comes from the library's deterministic generator (vksift_ext_genSyntheticImage, host code, no GPU)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from vulkansift_amd import api
    img = api.gen_synthetic_image(0x5EED0000, 640, 480)
    feats, counts = detect(img)
    np.save(os.path.join(HERE, "feats_c2_frame0_np.npy"), feats)
    print("c2 frame 0", counts)


def main():
    if "--bench-frame" in sys.argv:
        return bench_frame()
    img = image()
    np.save(os.path.join(HERE, "img_192x144.npy"), img)
    for name, kw in CONFIGS.items():
        feats, counts = detect(img, **kw)
        np.save(os.path.join(HERE, f"feats_192x144_{name}_np.npy"), feats)
        print(name, counts)


if __name__ == "__main__":
    main()
