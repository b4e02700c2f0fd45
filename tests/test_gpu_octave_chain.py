"""vksift_hip_octave_chain (the trailing octaves that fit the LDS, built by one launch): every plane it writes equals the oracle's
(blur_plane + nearest 2:1, oracle/sift_oracle.c) and the per-scale launches' (VKSIFT_LDS_CHAIN=0), and so do the features."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [
    (640, 480, {}),                                   # 1280x960 -> chain = 160x120, 80x60
    (640, 480, {"use_input_upsampling": False}),      # chain = 160x120, 80x60, 40x30
    (1280, 720, {}),                                  # 160x90, 80x45
    (320, 240, {}),                                   # 160x120 is octave 2
    (160, 120, {"use_input_upsampling": False}),      # octave 0 itself fits: the chain starts at octave 1
    (200, 136, {}),                                   # 400x272 -> 200x136 (too large), 100x68, 50x34 (not a multiple of 4: no chain)
    (512, 512, {"nb_scales_per_octave": 4}),          # 7 layers
    (512, 384, {"nb_scales_per_octave": 2, "seed_scale_sigma": 1.2}),  # other tap counts (the generic passes)
]


@pytest.mark.parametrize("w,h,kw", SHAPES)
def test_chain_planes_and_features_equal_oracle_and_per_scale_launches(vk, oracle, monkeypatch, w, h, kw):
    img = vk.gen_synthetic_image_family(9000 + w + h, w, h, (w // 8) % 3)
    okw = {k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()}
    pyr = oracle.Pyramid(oracle.default_config(math_mode=1, **okw), img)
    S = kw.get("nb_scales_per_octave", 3)
    out = {}
    for chain in ("0", "1"):
        monkeypatch.setenv("VKSIFT_LDS_CHAIN", chain)
        monkeypatch.setenv("VKSIFT_LDS_CHAIN_MAX", "19200")
        with vk.Instance(vk.default_config(input_image_max_size=w * h, **kw)) as inst:
            for rep in range(2):  # the second run replays the captured sequence where there is one
                inst.detectFeatures(img, 0)
                feats = inst.downloadFeatures(0)
            assert inst.getScaleSpaceNbOctaves() == pyr.nb_octaves
            planes = [[inst.downloadScaleSpaceImage(o, s) for s in range(S + 3)] for o in range(pyr.nb_octaves)]
        out[chain] = (feats, planes)
    for o in range(pyr.nb_octaves):
        for s in range(S + 3):
            ref = pyr.gauss(o, s)
            assert np.array_equal(out["1"][1][o][s].view(np.uint32), ref.view(np.uint32)), (o, s)
            assert np.array_equal(out["0"][1][o][s].view(np.uint32), ref.view(np.uint32)), (o, s)
    assert out["0"][0].tobytes() == out["1"][0].tobytes()
    ref_feats, _ = pyr.detect()
    assert out["1"][0].tobytes() == ref_feats.tobytes()


@pytest.mark.parametrize("w,h,kw", [SHAPES[0], SHAPES[1], SHAPES[6]])
def test_chain_that_declines_falls_back_to_trunk_and_branch_launches(vk, oracle, monkeypatch, w, h, kw):
    """VKSIFT_LDS_CHAIN=refuse: the host's pre-check passes, the chain launch shim declines (-1). In a forked detection the octaves it
    was to build then need BOTH halves of the per-scale schedule — the trunk up to scale S and the branch behind it (round 4 queued the
    trunk only: scales S+1, S+2 of those octaves were never built and the scan read stale planes)."""
    img = vk.gen_synthetic_image_family(9100 + w + h, w, h, 1)
    okw = {k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()}
    pyr = oracle.Pyramid(oracle.default_config(math_mode=1, **okw), img)
    S = kw.get("nb_scales_per_octave", 3)
    other = vk.gen_synthetic_image_family(5, w, h, 0)
    monkeypatch.setenv("VKSIFT_LDS_CHAIN", "refuse")
    monkeypatch.setenv("VKSIFT_LDS_CHAIN_MAX", "19200")
    with vk.Instance(vk.default_config(input_image_max_size=w * h, **kw)) as inst:
        inst.detectFeatures(other, 0)          # leaves ITS planes behind: stale scales would not equal the oracle's
        inst.downloadFeatures(0)
        for rep in range(2):
            inst.detectFeatures(img, 0)
            feats = inst.downloadFeatures(0)
        planes = [[inst.downloadScaleSpaceImage(o, s) for s in range(S + 3)] for o in range(pyr.nb_octaves)]
    for o in range(pyr.nb_octaves):
        for s in range(S + 3):
            assert np.array_equal(planes[o][s].view(np.uint32), pyr.gauss(o, s).view(np.uint32)), (o, s)
    ref_feats, _ = pyr.detect()
    assert feats.tobytes() == ref_feats.tobytes()


def test_chain_in_a_batch_equals_single_detections(vk, monkeypatch):
    """(forked scale-space and chain serve detections of at most VKSIFT_FORK_MAX_COUNT = 4 images; 12 take the per-scale launches)"""
    w, h = 640, 480
    imgs = [vk.gen_synthetic_image_family(77 + i, w, h, i % 3) for i in range(12)]
    res = {}
    for chain in ("0", "1"):
        monkeypatch.setenv("VKSIFT_LDS_CHAIN", chain)
        monkeypatch.setenv("VKSIFT_LDS_CHAIN_MAX", "19200")
        with vk.Instance(vk.default_config(sift_buffer_count=12, input_image_max_size=w * h), batch_capacity=12) as inst:
            inst.detectFeaturesBatch(imgs, 0)
            res[chain] = [inst.downloadFeatures(i).tobytes() for i in range(12)]
            inst.detectFeaturesBatch(imgs[8:12], 0)          # a batch small enough for the fork + the chain
            res[chain] += [inst.downloadFeatures(i).tobytes() for i in range(4)]
    assert res["0"] == res["1"] and all(len(r) > 0 for r in res["1"])
    assert res["1"][12:] == res["1"][8:12]


def test_forked_batch_with_grouped_upload_equals_single_detections(vk, monkeypatch):
    """a batch on a single-pyramid instance, large enough for the grouped upload: octave 0 is built group by group behind the copies;
    (until round 5 such a batch also forked its scales and chained the coarsest octaves: that is for <= 4 images now)"""
    w, h, n = 160, 120, 70
    imgs = [vk.gen_synthetic_image_family(400 + i, w, h, i % 3) for i in range(n)]
    monkeypatch.setenv("VKSIFT_PYR_PINGPONG", "0")
    with vk.Instance(vk.default_config(sift_buffer_count=n, input_image_max_size=w * h), batch_capacity=n) as inst:
        for rep in range(2):
            inst.detectFeaturesBatch(imgs, 0)
            batch = [inst.downloadFeatures(i).tobytes() for i in range(n)]
    monkeypatch.setenv("VKSIFT_FORK_SCALES", "0")
    monkeypatch.setenv("VKSIFT_LDS_CHAIN", "0")
    with vk.Instance(vk.default_config(input_image_max_size=w * h)) as inst:
        for i in (0, 1, 33, 69):
            inst.detectFeatures(imgs[i], 0)
            assert inst.downloadFeatures(0).tobytes() == batch[i] and len(batch[i]) > 0
