#!/usr/bin/env python3
"""Regenerate the golden fixtures in this directory.

The reference has no golden vectors of its own and cannot run here, so these
fixtures are produced by the CPU oracle (oracle/sift_oracle.c) in this container and pin it against
drift: inputs + expected outputs only (no reference source text).

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from vulkansift_amd import api  # noqa: E402  (host-side synthetic generator only; no GPU needed)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def main():
    img = api.gen_synthetic_image(0x601D, 160, 120)
    np.save(os.path.join(HERE, "img_160x120.npy"), img)
    meta = {}
    for mode, name in ((1, "det"), (0, "libm")):
        cfg = O.default_config(math_mode=mode)
        pyr = O.Pyramid(cfg, img)
        feats, counts = pyr.detect()
        np.save(os.path.join(HERE, f"feats_160x120_default_{name}.npy"), feats)
        meta[name] = {"counts": counts, "n": int(len(feats))}
        if mode == 1:
            planes = {}
            for o in range(pyr.nb_octaves):
                for s in range(6):
                    planes[f"g{o}_{s}"] = crc(pyr.gauss(o, s))
                for s in range(5):
                    planes[f"d{o}_{s}"] = crc(pyr.dog(o, s))
            meta["plane_crc32"] = planes
            meta["resolutions"] = [pyr.resolution(o) for o in range(pyr.nb_octaves)]
        pyr.close()
    # a second configuration: no up-sampling, VLFeat descriptor layout, unlimited orientations, direct taps
    cfg = O.default_config(math_mode=1, use_input_upsampling=0, use_vlfeat_format=1, max_nb_orientation_per_keypoint=0,
                           use_hardware_interpolated_blur=0)
    feats, counts = O.detect(cfg, img)
    np.save(os.path.join(HERE, "feats_160x120_noups_vlfeat_det.npy"), feats)
    meta["noups_vlfeat_det"] = {"counts": counts, "n": int(len(feats))}

    a = api.gen_synthetic_descriptors(0xA, 256)
    b = api.gen_synthetic_descriptors(0xB, 300)
    b[1] = b[0]          # quirk Q7 tie
    b[40] = b[17]        # duplicate rows
    a[5] = b[0]
    a[9] = b[40]
    np.save(os.path.join(HERE, "desc_a.npy"), a)
    np.save(os.path.join(HERE, "desc_b.npy"), b)
    np.save(os.path.join(HERE, "matches_a_b.npy"), O.match_2nn(a, b))
    json.dump(meta, open(os.path.join(HERE, "meta.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in meta.items() if k != "plane_crc32"}))


if __name__ == "__main__":
    main()
