"""End-to-end geometric sanity of detect -> match -> filter on a known homography (tests/quality.py)."""
import numpy as np
import pytest

import quality


def test_oracle_matches_follow_the_homography(oracle):
    """CPU oracle, small image: filtered matches must be geometrically correct"""
    from vulkansift_amd import api  # only for the synthetic image generator symbol (no GPU call)
    w, h = 256, 192
    img1 = api.gen_synthetic_image(31, w, h)
    H = quality.homography(w, h)
    img2 = quality.warp(img1, H)
    cfg = oracle.default_config(math_mode=0)
    f1, _ = oracle.detect(cfg, img1)
    f2, _ = oracle.detect(cfg, img2)
    assert len(f1) > 100 and len(f2) > 100
    m12, m21 = oracle.match_2nn(f1, f2), oracle.match_2nn(f2, f1)
    ia, ib = oracle.filter_matches(m12, m21, 0.75, True)
    s = quality.score(f1, f2, ia, ib, H, w, h)
    assert s["matches"] >= 40, s
    assert s["precision"] >= 0.9, s
    assert s["repeatability"] >= 0.4, s


@pytest.mark.gpu
def test_gpu_matches_follow_the_homography(vk):
    """HIP path at the benchmark resolution, through vksift_ext_matchFeaturesFiltered"""
    w, h = 640, 480
    img1 = vk.gen_synthetic_image(33, w, h)
    H = quality.homography(w, h)
    img2 = quality.warp(img1, H)
    with vk.Instance(vk.default_config()) as inst:
        inst.detectFeatures(img1, 0)
        inst.detectFeatures(img2, 1)
        inst.matchFeaturesFiltered([0], [1], 0.75, True)
        fm = inst.downloadFilteredMatches(0)
        f1, f2 = inst.downloadFeatures(0), inst.downloadFeatures(1)
    s = quality.score(f1, f2, fm["idx_a"], fm["idx_b"], H, w, h)
    assert s["matches"] >= 200, s
    assert s["precision"] >= 0.9, s
    assert s["repeatability"] >= 0.4, s
