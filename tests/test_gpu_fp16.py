"""VKSIFT_PYRAMID_PRECISION_FLOAT16 as this build defines it — scale-space and DoG images STORED as IEEE
binary16 (round to nearest even), widened exactly on every read, all arithmetic fp32 — against the oracle's pyramid_fp16
mode: byte-exact planes and features; and against the fp32 pipeline: the same keypoints within a stated tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F16 = 1  # VKSIFT_PYRAMID_PRECISION_FLOAT16


def _cfgs(vk, oracle, **kw):
    okw, vkw = {}, {}
    for k, v in kw.items():
        if k in ("use_input_upsampling", "use_hardware_interpolated_blur"):
            okw[k], vkw[k] = int(v), bool(v)
        elif k == "descriptor_format":
            okw["use_vlfeat_format"], vkw[k] = int(v), int(v)
        else:
            okw[k], vkw[k] = v, v
    return vk.default_config(pyramid_precision_mode=F16, **vkw), oracle.default_config(math_mode=1, pyramid_fp16=1, **okw)


@pytest.mark.parametrize("w,h,kw", [
    (320, 240, {}),
    (200, 150, {"use_input_upsampling": False}),
    (257, 131, {"use_hardware_interpolated_blur": False}),   # odd width: the generic tile kernel + separate blit / down-sampling launches
    (368, 224, {"nb_scales_per_octave": 2}),
    (100, 75, {}),
])
def test_fp16_planes_bit_exact(vk, oracle, w, h, kw):
    vcfg, ocfg = _cfgs(vk, oracle, **kw)
    img = vk.gen_synthetic_image(17, w, h)
    with vk.Instance(vcfg) as inst:
        inst.detectFeatures(img, 0)
        pyr = oracle.Pyramid(ocfg, img)
        assert inst.getScaleSpaceNbOctaves() == pyr.nb_octaves
        S = vcfg.nb_scales_per_octave
        for o in range(pyr.nb_octaves):
            for s in range(S + 3):
                g, ref = inst.downloadScaleSpaceImage(o, s), pyr.gauss(o, s)
                assert np.array_equal(g.view(np.uint32), ref.view(np.uint32)), ("gauss", o, s, np.abs(g - ref).max())
                assert np.array_equal(g, g.astype(np.float16).astype(np.float32))      # really binary16 values
            for s in range(S + 2):
                d, ref = inst.downloadDoGImage(o, s), pyr.dog(o, s)
                assert np.array_equal(d.view(np.uint32), ref.view(np.uint32)), ("dog", o, s, np.abs(d - ref).max())


@pytest.mark.parametrize("w,h,kw", [
    (320, 240, {}),
    (200, 150, {"use_input_upsampling": False}),
    (320, 240, {"descriptor_format": 1, "max_nb_orientation_per_keypoint": 0}),
    (640, 480, {}),
])
def test_fp16_features_bit_exact(vk, oracle, w, h, kw):
    vcfg, ocfg = _cfgs(vk, oracle, **kw)
    img = vk.gen_synthetic_image(19, w, h)
    with vk.Instance(vcfg) as inst:
        inst.detectFeatures(img, 0)
        feats = inst.downloadFeatures(0)
    ref, _ = oracle.detect(ocfg, img)
    assert len(ref) > 50
    assert len(feats) == len(ref), (len(feats), len(ref))
    assert feats.tobytes() == ref.tobytes()


def test_fp16_batch_equals_single_and_1080p(vk, oracle):
    imgs = [vk.gen_synthetic_image(2000 + i, 256, 192) for i in range(9)]
    vcfg, ocfg = _cfgs(vk, oracle)
    vcfg.sift_buffer_count = 9
    with vk.Instance(vcfg, batch_capacity=9) as inst:
        inst.detectFeaturesBatch(imgs, 0)
        got = [inst.downloadFeatures(i) for i in range(9)]
    for i in (0, 4, 8):
        assert got[i].tobytes() == oracle.detect(ocfg, imgs[i])[0].tobytes(), i
    w, h = 1920, 1080
    img = vk.gen_synthetic_image(77, w, h)
    vcfg, ocfg = _cfgs(vk, oracle, input_image_max_size=w * h)
    with vk.Instance(vcfg) as inst:
        inst.detectFeatures(img, 0)
        feats = inst.downloadFeatures(0)
    ref, _ = oracle.detect(ocfg, img)
    assert len(ref) > 3000 and feats.tobytes() == ref.tobytes()


def test_fp16_against_fp32_pipeline(vk):
    """what the storage format costs: binary16 texels quantise the planes to ~5e-4 at mid grey (5 % of the DoG pre-filter
    threshold), so marginal extrema come and go; the keypoints both pipelines find agree closely"""
    img = vk.gen_synthetic_image(108, 640, 480)
    out = {}
    for mode in (0, F16):
        with vk.Instance(vk.default_config(pyramid_precision_mode=mode)) as inst:
            inst.detectFeatures(img, 0)
            out[mode] = inst.downloadFeatures(0)
    a, b = out[0], out[F16]
    assert 0.6 * len(a) < len(b) < 1.4 * len(a), (len(a), len(b))
    # nearest fp32 keypoint (same octave and scale) of every fp16 keypoint
    hit = 0
    dpos, dori, drms = [], [], []
    for f in b:
        m = (a["octave_idx"] == f["octave_idx"]) & (a["scale_idx"] == f["scale_idx"])
        if not m.any():
            continue
        c = a[m]
        d = np.hypot(c["scale_x"] - f["scale_x"], c["scale_y"] - f["scale_y"])
        dth = np.abs(((c["orientation"] - f["orientation"] + np.pi) % (2 * np.pi)) - np.pi)
        k = np.argmin(d + dth)
        if d[k] < 0.5 and dth[k] < 0.3:
            hit += 1
            dpos.append(d[k]), dori.append(dth[k])
            drms.append(np.sqrt(((c["descriptor"][k].astype(float) - f["descriptor"].astype(float)) ** 2).sum()) / 512.0)
    assert hit > 0.7 * len(b), (hit, len(a), len(b))
    assert np.median(dpos) < 0.05 and np.median(dori) < 0.02 and np.median(drms) < 0.05, (np.median(dpos), np.median(dori), np.median(drms))
