"""The oracle against the committed golden fixtures (tests/golden/, made by make_golden.py), and the
distance between its two math back-ends (libm vs the deterministic detmath.h used on the GPU)."""
import json
import os
import zlib

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def _meta():
    return json.load(open(os.path.join(G, "meta.json")))


def _same_features(a, b):
    assert len(a) == len(b)
    for name in a.dtype.names:
        assert np.array_equal(a[name], b[name]) or (a[name].dtype.kind == "f" and np.array_equal(a[name].view(np.uint32), b[name].view(np.uint32))), name


def test_det_mode_reproduces_golden(oracle):
    img = _load("img_160x120.npy")
    meta = _meta()
    pyr = oracle.Pyramid(oracle.default_config(math_mode=1), img)
    assert [list(pyr.resolution(o)) for o in range(pyr.nb_octaves)] == meta["resolutions"]
    for o in range(pyr.nb_octaves):
        for s in range(6):
            assert zlib.crc32(pyr.gauss(o, s).tobytes()) == meta["plane_crc32"][f"g{o}_{s}"]
        for s in range(5):
            assert zlib.crc32(pyr.dog(o, s).tobytes()) == meta["plane_crc32"][f"d{o}_{s}"]
    feats, counts = pyr.detect()
    assert counts == meta["det"]["counts"]
    _same_features(feats, _load("feats_160x120_default_det.npy"))


def test_second_config_reproduces_golden(oracle):
    img = _load("img_160x120.npy")
    cfg = oracle.default_config(math_mode=1, use_input_upsampling=0, use_vlfeat_format=1, max_nb_orientation_per_keypoint=0,
                                use_hardware_interpolated_blur=0)
    feats, counts = oracle.detect(cfg, img)
    assert counts == _meta()["noups_vlfeat_det"]["counts"]
    _same_features(feats, _load("feats_160x120_noups_vlfeat_det.npy"))


def test_matcher_reproduces_golden(oracle):
    got = oracle.match_2nn(_load("desc_a.npy"), _load("desc_b.npy"))
    ref = _load("matches_a_b.npy")
    for name in ref.dtype.names:
        assert np.array_equal(got[name], ref[name]), name
    assert ref["idx_b1"][5] == 1 and ref["idx_b2"][5] == 0      # Q7
    assert ref["idx_b1"][9] == 17 and ref["idx_b2"][9] == 40    # duplicate rows: earlier index first


def test_libm_and_det_math_agree_within_tolerance(oracle):
    """Any conforming exp/atan2/sin/cos may move a result by at most this much (the stated float tolerance
    of BASELINE.json: keypoints within 1e-4 px / 1e-5 rel sigma, descriptors < 1e-3 RMS of the 512-norm)."""
    det = _load("feats_160x120_default_det.npy")
    libm = _load("feats_160x120_default_libm.npy")
    assert len(det) == len(libm)
    assert np.array_equal(det["scale_idx"], libm["scale_idx"]) and np.array_equal(det["octave_idx"], libm["octave_idx"])
    for name in ("x", "y", "scale_x", "scale_y", "intensity"):
        assert np.array_equal(det[name], libm[name]), name  # no transcendental involved
    assert np.abs(det["sigma"] / libm["sigma"] - 1).max() < 1e-6
    assert np.abs(det["orientation"] - libm["orientation"]).max() < 1e-5
    diff = det["descriptor"].astype(np.float64) - libm["descriptor"].astype(np.float64)
    rms = np.sqrt((diff ** 2).mean(axis=1)) / 512.0
    assert rms.max() < 1e-3, rms.max()
    assert (np.abs(diff) <= 1).mean() > 0.999


def test_descriptor_invariants(oracle):
    feats = _load("feats_160x120_default_det.npy")
    norms = np.sqrt((feats["descriptor"].astype(np.float64) ** 2).sum(1))
    assert np.all((norms > 480) & (norms < 520))          # ~512 after the final scaling, truncation loses a little
    # bin 35 with the uint-wrap interpolation of quirk Q3 lands exactly on 2*pi (not wrapped by the reference)
    assert np.all((feats["orientation"] >= 0) & (feats["orientation"] <= 2 * np.pi + 1e-5))
    m = oracle.match_2nn(feats, feats)
    uniq = np.unique(feats["descriptor"], axis=0).shape[0] == len(feats)
    if uniq:
        assert np.array_equal(m["idx_b1"], m["idx_a"]) and np.all(m["dist_a_b1"] == 0)


def test_ubc_and_vlfeat_descriptors_are_bin_permutations(oracle):
    """quirk Q5 invariant: the two formats differ by the orientation-bin direction only (bin k <-> (8-k) % 8),
    up to +-1 quantisation from the different trilinear split."""
    img = _load("img_160x120.npy")
    a, _ = oracle.detect(oracle.default_config(math_mode=1, use_vlfeat_format=0), img)
    b, _ = oracle.detect(oracle.default_config(math_mode=1, use_vlfeat_format=1), img)
    assert len(a) == len(b)
    da = a["descriptor"].reshape(-1, 16, 8).astype(int)
    db = b["descriptor"].reshape(-1, 16, 8).astype(int)
    perm = [(8 - k) % 8 for k in range(8)]
    # energy per spatial cell is format independent
    assert np.abs(da.sum(2) - db.sum(2)).mean() < 3.0
    close = np.abs(da - db[:, :, perm]) <= 40
    assert close.mean() > 0.9
