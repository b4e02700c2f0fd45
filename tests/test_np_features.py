"""The C oracle (libm math) against the independent numpy restatement of K4 / K5 / K6 (tests/np_features.py).

Tolerances and why they are not zero: the numpy side solves the refinement in float64 and evaluates exp / atan2 in
float64 rounded once to fp32, the oracle works in fp32 with libm. Values agree to the last bits; the fixed-point
truncations uint(x) can then differ by one count per contribution, and a final descriptor byte by one. Measured on
these images: sub-pixel positions differ by at most 2.4e-7 px, smoothed orientation histograms and angles are
identical, raw descriptor accumulators differ by at most one count, descriptor bytes are identical."""
import numpy as np
import pytest

import np_features as NF


def _img(seed, w, h):
    """dense blob field + noise: a few hundred keypoints at 160x120"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.full((h, w), 128.0)
    for _ in range(int(w * h / 55)):
        cx, cy, s = rng.uniform(0, w), rng.uniform(0, h), np.exp(rng.uniform(np.log(0.8), np.log(7)))
        a = rng.uniform(25, 100) * rng.choice([-1, 1])
        img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    img += rng.uniform(-4, 4, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


CONFIGS = [
    dict(),
    dict(use_input_upsampling=0),
    dict(use_vlfeat_format=1, max_nb_orientation_per_keypoint=0),
    dict(nb_scales_per_octave=2, use_hardware_interpolated_blur=0),
    dict(nb_scales_per_octave=5, intensity_threshold=0.03, edge_threshold=6.0),
]


def _octave_arrays(pyr, o):
    S = pyr.S
    return np.stack([pyr.gauss(o, s) for s in range(S + 3)]), np.stack([pyr.dog(o, s) for s in range(S + 2)])


def _key(f):
    return (int(f["scale_idx"]), float(f["scale_y"]), float(f["scale_x"]))


@pytest.mark.parametrize("overrides", CONFIGS)
def test_extract_keypoints_same_set_and_values(oracle, overrides):
    img = _img(11, 160, 120)
    cfg = oracle.default_config(math_mode=0, **overrides)
    pyr = oracle.Pyramid(cfg, img)
    S = cfg.nb_scales_per_octave
    total = 0
    for o in range(pyr.nb_octaves):
        _, dog = _octave_arrays(pyr, o)
        ref, n = pyr.extract_keypoints(o)
        assert n == len(ref)
        got = NF.extract_keypoints(dog, S, o - cfg.use_input_upsampling, cfg.seed_scale_sigma, cfg.intensity_threshold, cfg.edge_threshold)
        assert len(got) == len(ref), (o, len(got), len(ref))
        total += len(ref)
        # both are in raster order of the *starting* texel; compare record by record
        for name, tol in (("scale_x", 4e-5), ("scale_y", 4e-5), ("x", 8e-5), ("y", 8e-5), ("intensity", 1e-7)):
            assert np.abs(got[name] - ref[name]).max(initial=0) < tol, (o, name)
        assert np.abs(got["sigma"] / ref["sigma"] - 1).max(initial=0) < 1e-6
        assert np.array_equal(got["scale_idx"], ref["scale_idx"]) and np.array_equal(got["octave_idx"], ref["octave_idx"])
    assert total > 60


def test_q1_refinement_reaches_the_missing_layer(oracle):
    """quirk Q1 is exercised: some accepted keypoint ends on DoG layer S+1 (its s+1 neighbour is the zero plane)"""
    hits = 0
    for seed in (11, 12, 13, 14):
        img = _img(seed, 200, 150)
        cfg = oracle.default_config(math_mode=0)
        pyr = oracle.Pyramid(cfg, img)
        for o in range(pyr.nb_octaves):
            _, dog = _octave_arrays(pyr, o)
            ref, _ = pyr.extract_keypoints(o)
            got = NF.extract_keypoints(dog, 3, o - 1)
            assert len(got) == len(ref)
            hits += int((got["scale_idx"] >= 4).sum())
            assert np.array_equal(got["scale_idx"], ref["scale_idx"])
    assert hits >= 1


@pytest.mark.parametrize("overrides", CONFIGS[:3])
def test_orientation_histograms_and_angles(oracle, overrides):
    img = _img(21, 160, 120)
    cfg = oracle.default_config(math_mode=0, **overrides)
    pyr = oracle.Pyramid(cfg, img)
    n_kp = n_multi = n_border = 0
    for o in range(pyr.nb_octaves):
        gauss, _ = _octave_arrays(pyr, o)
        W, H = pyr.resolution(o)
        kps, _ = pyr.extract_keypoints(o)
        for kp in kps:
            ang_ref, hist_ref = pyr.orientations(o, kp)
            ang, hist = NF.orientations(gauss[int(kp["scale_idx"])], kp, max_nb_orientation=0)
            # each of the <= (2r+1)^2 contributions may truncate one count differently; smoothing keeps that scale
            assert np.abs(hist.astype(np.int64) - hist_ref.astype(np.int64)).max() <= 16, (o, kp["scale_x"], kp["scale_y"])
            assert len(ang) == len(ang_ref)
            assert np.abs(ang - ang_ref).max(initial=0) < 1e-6
            n_kp += 1
            n_multi += len(ang) > 1
            r = int(np.floor(4.5 * kp["sigma"] / 2.0 ** int(kp["octave_idx"])))
            n_border += (kp["scale_x"] < r) or (kp["scale_y"] < r) or (kp["scale_x"] > W - r) or (kp["scale_y"] > H - r)
    assert n_kp > 60 and n_multi > 3 and n_border > 3      # Q2's out-of-image window texels are exercised


def test_q3_wrapped_peak_interpolation_values():
    """the uint wrap-around formula on hand-made histograms: h[p] < h[n] gives a huge numerator, not a negative one"""
    # reproduce the formula through NF.orientations' own arithmetic on a synthetic histogram
    h = np.zeros(36, np.uint32)
    h[10], h[9], h[11] = 1000, 400, 600          # p < n
    prev, nxt = np.roll(h, 1), np.roll(h, -1)
    with np.errstate(over="ignore"):
        num = np.uint32(prev[10] - nxt[10])
        den = np.uint32(np.uint32(prev[10] - np.uint32(2) * h[10]) + nxt[10])
    assert int(num) == 2 ** 32 - 200 and int(den) == 2 ** 32 - 1000
    pos = np.float32(10) + np.float32(0.5) * (np.float32(num) / np.float32(den))
    assert abs(float(pos) - 10.5) < 1e-6         # not 10 - 0.1 as the textbook parabola would give


@pytest.mark.parametrize("overrides", CONFIGS[:4])
def test_descriptors(oracle, overrides):
    img = _img(31, 160, 120)
    cfg = oracle.default_config(math_mode=0, **overrides)
    pyr = oracle.Pyramid(cfg, img)
    feats, counts = pyr.detect()
    start = 0
    n = n_byte_off = 0
    worst_rms = 0.0
    for o in range(pyr.nb_octaves):
        gauss, _ = _octave_arrays(pyr, o)
        sec = feats[start:start + counts[o]]
        start += counts[o]
        for f in sec[::2]:
            desc_ref, raw_ref = pyr.descriptor(o, f)
            assert np.array_equal(desc_ref, f["descriptor"])
            desc, raw = NF.descriptor(gauss[int(f["scale_idx"])], f, f["orientation"], vlfeat=bool(cfg.use_vlfeat_format))
            # raw accumulators: every contribution may truncate one count differently
            assert np.abs(raw.astype(np.int64) - raw_ref.astype(np.int64)).max() <= 4
            d = desc.astype(np.int32) - desc_ref.astype(np.int32)
            assert np.abs(d).max() <= 1
            n_byte_off += int((d != 0).sum())
            worst_rms = max(worst_rms, float(np.sqrt((d.astype(np.float64) ** 2).sum()) / 512.0))
            n += 1
    assert n > 30
    assert worst_rms < 1e-3 * 4 and n_byte_off <= 0.002 * n * 128


def test_ubc_is_a_bin_permutation_of_vlfeat(oracle):
    """Q5 (floored % on negative bins): the UBC descriptor is the VLFeat one with orientation bin k -> (8 - k) % 8"""
    img = _img(41, 128, 96)
    cfg = oracle.default_config(math_mode=0)
    pyr = oracle.Pyramid(cfg, img)
    gauss, dog = _octave_arrays(pyr, 1)
    kps = NF.extract_keypoints(dog, 3, 0)
    assert len(kps) > 10
    perm = np.array([(8 - k) % 8 for k in range(8)])
    for kp in kps[:12]:
        ang, _ = NF.orientations(gauss[int(kp["scale_idx"])], kp)
        th = ang[0] if len(ang) else np.float32(0)
        _, raw_u = NF.descriptor(gauss[int(kp["scale_idx"])], kp, th, vlfeat=False)
        _, raw_v = NF.descriptor(gauss[int(kp["scale_idx"])], kp, th, vlfeat=True)
        ru = raw_u.reshape(16, 8).astype(np.int64)
        rv = raw_v.reshape(16, 8).astype(np.int64)
        # the fractional bin weight differs between the two roundings: compare adjacent-bin sums loosely, support exactly
        assert np.abs(ru[:, perm].sum(1) - rv.sum(1)).max() <= 64
        assert np.abs(np.roll(ru[:, perm], 1, axis=1) - rv).max() <= 0.02 * rv.max() + 64 or np.abs(ru[:, perm] - rv).max() <= 0.02 * rv.max() + 64


@pytest.mark.parametrize("overrides", CONFIGS[:3])
def test_full_octave_chain_equals_oracle_detect(oracle, overrides):
    """numpy K4 -> K5 -> K6 chained on the oracle's planes against orc_detect: same features in the same order"""
    img = _img(51, 128, 96)
    cfg = oracle.default_config(math_mode=0, **overrides)
    pyr = oracle.Pyramid(cfg, img)
    feats, counts = pyr.detect()
    start = 0
    for o in range(pyr.nb_octaves):
        gauss, dog = _octave_arrays(pyr, o)
        ref = feats[start:start + counts[o]]
        start += counts[o]
        got = NF.detect_octave(gauss, dog, cfg.nb_scales_per_octave, o - cfg.use_input_upsampling, cfg.seed_scale_sigma, cfg.intensity_threshold,
                               cfg.edge_threshold, cfg.max_nb_orientation_per_keypoint, bool(cfg.use_vlfeat_format))
        assert len(got) == len(ref), (o, len(got), len(ref))
        assert np.abs(got["scale_x"] - ref["scale_x"]).max(initial=0) < 4e-5
        assert np.abs(got["orientation"] - ref["orientation"]).max(initial=0) < 1e-6
        d = got["descriptor"].astype(np.int32) - ref["descriptor"].astype(np.int32)
        assert np.abs(d).max(initial=0) <= 1
