"""How far does an 8-bit-weight texture sampler move the result? Oracle with exact bilinear taps (what the HIP kernels compute) against the
oracle's sampler model (orc_Config.sampler_model) on the bench frame and two 1080p frames. usage: sampler_gap.py [--fast]"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle import oracle as O
from vulkansift_amd import api


def compare(img, **kw):
    a, _ = O.detect(O.default_config(math_mode=0, max_nb_sift_per_buffer=200000, **kw), img)
    b, _ = O.detect(O.default_config(math_mode=0, max_nb_sift_per_buffer=200000, sampler_model=1, **kw), img)
    # pair features: same octave and scale index, nearest position within 1 octave-0 pixel, orientation within 0.2 rad
    from collections import defaultdict
    cells = defaultdict(list)
    for j, f in enumerate(b):
        cells[(int(f["octave_idx"]), int(f["scale_idx"]), int(f["x"] // 4), int(f["y"] // 4))].append(j)
    used = set()
    pairs = []
    for i, f in enumerate(a):
        best, bd = -1, 1.0
        cx, cy = int(f["x"] // 4), int(f["y"] // 4)
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for j in cells.get((int(f["octave_idx"]), int(f["scale_idx"]), cx + dx, cy + dy), ()):
                    if j in used:
                        continue
                    g = b[j]
                    d = np.hypot(f["x"] - g["x"], f["y"] - g["y"])
                    da = abs((f["orientation"] - g["orientation"] + np.pi) % (2 * np.pi) - np.pi)
                    if d < bd and da < 0.2:
                        best, bd = j, d
        if best >= 0:
            used.add(best)
            pairs.append((i, best))
    ia = np.array([p[0] for p in pairs]); ib = np.array([p[1] for p in pairs])
    pos = np.hypot(a["x"][ia] - b["x"][ib], a["y"][ia] - b["y"][ib])
    dd = a["descriptor"][ia].astype(np.float64) - b["descriptor"][ib].astype(np.float64)
    rms = np.sqrt((dd ** 2).sum(axis=1)) / 512.0
    return {"features_exact": len(a), "features_sampler": len(b), "paired": len(pairs), "unpaired_frac": 1.0 - 2.0 * len(pairs) / (len(a) + len(b)),
            "pos_rms_px": float(np.sqrt((pos ** 2).mean())), "pos_max_px": float(pos.max()), "desc_rms_of_norm_mean": float(rms.mean()),
            "desc_rms_of_norm_p99": float(np.quantile(rms, 0.99)), "desc_rms_of_norm_max": float(rms.max()),
            "desc_per_element_rms_over_512_median": float(np.median(rms / np.sqrt(128.0))), "desc_per_element_rms_over_512_p99": float(np.quantile(rms / np.sqrt(128.0), 0.99)),
            "sigma_rel_max": float(np.abs(a["sigma"][ia] / b["sigma"][ib] - 1).max())}


if __name__ == "__main__":
    cases = [("C2 frame 0 (640x480)", api.gen_synthetic_image(0x5EED0000, 640, 480))]
    if "--fast" not in sys.argv:
        cases += [("1080p frame 0", api.gen_synthetic_image(0x5EED0000, 1920, 1080)), ("1080p edges family", api.gen_synthetic_image_family(3, 1920, 1080, api.SYNTH_EDGES))]
    for name, img in cases:
        print(name, compare(img))
