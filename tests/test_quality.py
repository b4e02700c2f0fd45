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


@pytest.mark.gpu
def test_four_metrics_over_five_warps_hip_and_oracle_side_by_side(vk, oracle):
    """perf_matching.cpp's four metrics over five warps of increasing difficulty: the HIP path (GPU-side cross-check + ratio test)
    and the oracle (CPU filter of perf_common.cpp:123-169) must report IDENTICAL numbers — and sane ones"""
    w, h = 400, 300
    img1 = vk.gen_synthetic_image(35, w, h)
    ocfg = oracle.default_config(math_mode=1)
    o1, _ = oracle.detect(ocfg, img1)
    rows = []
    with vk.Instance(vk.default_config()) as inst:
        for k, kw in enumerate(quality.WARPS):
            H = quality.homography(w, h, **kw)
            img2 = quality.warp(img1, H)
            inst.detectFeatures(img1, 0)
            inst.detectFeatures(img2, 1)
            inst.matchFeaturesFiltered([0], [1], 0.75, True)
            fm = inst.downloadFilteredMatches(0)
            f1, f2 = inst.downloadFeatures(0), inst.downloadFeatures(1)
            s_hip = quality.score(f1, f2, fm["idx_a"], fm["idx_b"], H, w, h)
            o2, _ = oracle.detect(ocfg, img2)
            ia, ib = oracle.filter_matches(oracle.match_2nn(o1, o2), oracle.match_2nn(o2, o1), 0.75, True)
            s_orc = quality.score(o1, o2, ia, ib, H, w, h)
            assert s_hip == s_orc, (k, s_hip, s_orc)
            rows.append(s_hip)
    # easy warps: most keypoints re-detected and matched correctly; the hardest (70 degrees, x1.6, strong perspective) still finds geometry
    assert rows[0]["precision"] > 0.95 and rows[0]["matching_score"] > 0.3 and rows[0]["repeatability"] > 0.6, rows[0]
    assert rows[1]["precision"] > 0.9 and rows[1]["putative_match_ratio"] > 0.15, rows[1]
    assert all(r["precision"] > 0.8 for r in rows[:4]), rows
    assert rows[4]["matches"] >= 10 and rows[4]["precision"] > 0.5, rows[4]
    assert rows[0]["matching_score"] > rows[2]["matching_score"] > rows[4]["matching_score"]
