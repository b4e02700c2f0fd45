"""k_blur_wide (four texels per lane, 256-column strips: pyramid.hip) against k_blur_lean (two texels per lane) and the oracle, and the
measured placement of the scale-space buffers (vksift_instance.c: place_pyramid_buffers).

The wide form is what the 9-, 11- and 13-tap launches of eligible widths take by default, so every plane comparison of
test_gpu_parity.py / test_gpu_configs.py already runs it against the oracle; here both forms are forced through the development knob
vksift_hip_tune(VKSIFT_TUNE_WIDE_MASK, ...) on the same images — every tap count the wide kernel is instantiated for (5, 7, 9, 11, 13),
widths with a partly filled last strip, the fused down-sampling store, image edges inside the first / last strip — and every plane and
feature has to be the same bit for bit. (The first such comparison found a hardware hazard the compiler does not cover: a 16-byte
buffer store with a scalar offset followed at once by a write of its data registers, pyramid.hip: store_b128_stream.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [
    (640, 480, {}),                                                    # octave 0: 1280x960 (5 strips), octave 1: 640 wide (lean: 128 idle columns)
    (960, 540, {}),                                                    # 1920x1080: 8 strips, the last one half full
    (1920, 1080, {"use_input_upsampling": False}),                     # the same plane sizes from a plane source, 6 octaves
    (1000, 600, {"use_hardware_interpolated_blur": False}),            # 2000 wide: 8 strips, 48 idle columns; even tap counts (6, 8, 14 -> lean) beside 9 / 11
    (768, 512, {"nb_scales_per_octave": 4}),                           # 1536 = 6 strips exactly; 7 layers: other tap counts
    (512, 384, {"nb_scales_per_octave": 2, "seed_scale_sigma": 1.2}),  # 1024 wide, small tap counts (5 / 7 only with the mask forced)
]


@pytest.mark.parametrize("w,h,kw", SHAPES)
def test_wide_and_lean_blur_launches_give_identical_planes_and_features(vk, oracle, w, h, kw):
    img = vk.gen_synthetic_image_family(4200 + w + h, w, h, (w // 64) % 3)
    S = kw.get("nb_scales_per_octave", 3)
    L = vk.lib()
    out = {}
    try:
        for name, mask in (("lean", 0), ("wide", 0xFFFFF)):
            L.vksift_hip_tune(1, mask)
            with vk.Instance(vk.default_config(input_image_max_size=w * h, **kw)) as inst:
                inst.detectFeatures(img, 0)
                feats = inst.downloadFeatures(0)
                n_oct = inst.getScaleSpaceNbOctaves()
                planes = [[inst.downloadScaleSpaceImage(o, s) for s in range(S + 3)] for o in range(n_oct)]
            out[name] = (feats, planes)
    finally:
        L.vksift_hip_tune(1, -1)
    assert len(out["wide"][0]) > 50
    assert out["wide"][0].tobytes() == out["lean"][0].tobytes()
    for o, (pw, pl) in enumerate(zip(out["wide"][1], out["lean"][1])):
        for s, (a, b) in enumerate(zip(pw, pl)):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (o, s)
    okw = {k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()}
    ref, _ = oracle.detect(oracle.default_config(math_mode=1, **okw), img)
    assert out["wide"][0].tobytes() == ref.tobytes()


def test_wide_blur_in_a_batch_with_both_march_directions(vk):
    """a batch of 24 frames on a two-buffer instance: consecutive launches walk the batch in opposite directions (Plane::reverse), the grid
    is a multiple of 8 (XCD remap) — against the single-image results"""
    w, h, n = 640, 480, 24
    imgs = [vk.gen_synthetic_image_family(880 + i, w, h, i % 3) for i in range(n)]
    L = vk.lib()
    res = {}
    try:
        for name, mask in (("lean", 0), ("wide", 0xFFFFF)):
            L.vksift_hip_tune(1, mask)
            with vk.Instance(vk.default_config(sift_buffer_count=n, input_image_max_size=w * h), batch_capacity=n) as inst:
                for rep in range(2):
                    inst.detectFeaturesBatch(imgs, 0)
                res[name] = [inst.downloadFeatures(i).tobytes() for i in range(n)]
    finally:
        L.vksift_hip_tune(1, -1)
    assert res["wide"] == res["lean"] and all(len(r) > 164 * 100 for r in res["wide"])


PAIR_SHAPES = [
    (640, 480, {}),                                    # 1280 wide: 6 strips of 240 owned columns (the last one a third full); 640, 320 wide below
    (960, 540, {}),                                    # 1920 = 8 strips exactly
    (1000, 600, {}),                                   # 2000 wide: 9 strips, image edge 80 columns into the last one
    (126, 250, {}),                                    # 252 wide: one strip, both image edges inside it; octave 1 falls to the two-texel form
    (1920, 1080, {"use_input_upsampling": False}),
]


@pytest.mark.parametrize("w,h,kw", PAIR_SHAPES)
def test_two_scale_launch_in_both_lane_widths(vk, oracle, w, h, kw):
    """k_blur_pair_wide (four texels per lane) against k_blur_pair (two) through VKSIFT_TUNE_PAIR_FORM, and against the oracle"""
    img = vk.gen_synthetic_image_family(5100 + w + h, w, h, (w // 64) % 3)
    L = vk.lib()
    out = {}
    try:
        for name, form in (("two", 1), ("four", 2)):
            L.vksift_hip_tune(5, form)
            with vk.Instance(vk.default_config(input_image_max_size=w * h, **kw)) as inst:
                inst.detectFeatures(img, 0)
                feats = inst.downloadFeatures(0)
                n_oct = inst.getScaleSpaceNbOctaves()
                planes = [[inst.downloadScaleSpaceImage(o, s) for s in range(6)] for o in range(n_oct)]
            out[name] = (feats, planes)
    finally:
        L.vksift_hip_tune(5, 0)
    assert len(out["four"][0]) > 20
    for o, (pw, pl) in enumerate(zip(out["four"][1], out["two"][1])):
        for s, (a, b) in enumerate(zip(pw, pl)):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (o, s)
    assert out["four"][0].tobytes() == out["two"][0].tobytes()
    okw = {k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()}
    ref, _ = oracle.detect(oracle.default_config(math_mode=1, **okw), img)
    assert out["four"][0].tobytes() == ref.tobytes()


def test_two_scale_launch_widths_in_a_batch(vk):
    """24 frames on a two-buffer instance (both march directions, XCD remap), both lane widths, against each other"""
    w, h, n = 640, 480, 24
    imgs = [vk.gen_synthetic_image_family(1880 + i, w, h, i % 3) for i in range(n)]
    L = vk.lib()
    res = {}
    try:
        for name, form in (("two", 1), ("four", 2)):
            L.vksift_hip_tune(5, form)
            with vk.Instance(vk.default_config(sift_buffer_count=n, input_image_max_size=w * h), batch_capacity=n) as inst:
                for rep in range(2):
                    inst.detectFeaturesBatch(imgs, 0)
                res[name] = [inst.downloadFeatures(i).tobytes() for i in range(n)]
    finally:
        L.vksift_hip_tune(5, 0)
    assert res["four"] == res["two"] and all(len(r) > 164 * 100 for r in res["four"])


def test_scale_space_placement_is_measured_and_changes_no_result(vk, monkeypatch):
    """batch instances whose scale-space is large enough time candidate memory ranges and keep the fastest (vksift_ext_getScaleSpacePlacement
    reports the rates); VKSIFT_PYR_PLACEMENT=0 allocates plainly. Same features either way."""
    w, h, n = 640, 480, 16
    imgs = [vk.gen_synthetic_image_family(300 + i, w, h, i % 3) for i in range(n)]
    res = {}
    for mode in ("0", "5"):
        monkeypatch.setenv("VKSIFT_PYR_PLACEMENT", mode)
        with vk.Instance(vk.default_config(sift_buffer_count=n, input_image_max_size=w * h), batch_capacity=n) as inst:
            pl = inst.getScaleSpacePlacement()
            inst.detectFeaturesBatch(imgs, 0)
            res[mode] = [inst.downloadFeatures(i).tobytes() for i in range(n)]
        if mode == "0":
            assert pl["gbps"] == [] and pl["chosen"] == []
        else:
            # 16 x 51.75 MB = 828 MB: searched; the instance's one scale-space buffer is the fastest candidate, rates are plausible
            assert 2 <= len(pl["gbps"]) <= 5 and all(200.0 < g < 8000.0 for g in pl["gbps"])
            assert len(pl["chosen"]) == 1 and pl["chosen"][0] < len(pl["gbps"])
            assert pl["gbps"][pl["chosen"][0]] >= max(pl["gbps"]) - 1e-3
    assert res["0"] == res["5"]
    # two scale-space buffers (the mode of rounds 2-5): the two fastest candidates, the same features
    monkeypatch.setenv("VKSIFT_PYR_PLACEMENT", "5")
    monkeypatch.setenv("VKSIFT_PYR_PINGPONG", "2")
    with vk.Instance(vk.default_config(sift_buffer_count=n, input_image_max_size=w * h), batch_capacity=n) as inst:
        pl = inst.getScaleSpacePlacement()
        for rep in range(3):
            inst.detectFeaturesBatch(imgs, 0)
        assert [inst.downloadFeatures(i).tobytes() for i in range(n)] == res["0"]
    assert len(pl["chosen"]) == 2 and min(pl["gbps"][c] for c in pl["chosen"]) >= sorted(pl["gbps"])[-2] - 1e-3
    monkeypatch.delenv("VKSIFT_PYR_PINGPONG")
    # a single-image instance never searches
    with vk.Instance(vk.default_config(input_image_max_size=w * h)) as inst:
        assert inst.getScaleSpacePlacement()["gbps"] == []


def test_page_locked_destinations_receive_the_same_records(vk):
    """vksift_ext_pinHostMemory: a page-locked destination gets the records of a batched detection / matching by DMA straight from the
    packed device copy (no pinned staging, no host memcpy) — byte for byte what the pageable path returns, in any interleaving of pinned
    and pageable downloads of the same detection"""
    w, h, n = 320, 240, 16
    imgs = [vk.gen_synthetic_image_family(610 + i, w, h, i % 3) for i in range(n)]
    L = vk.lib()
    cap = 20000
    cfg = vk.default_config(sift_buffer_count=n, input_image_max_size=w * h, max_nb_sift_per_buffer=cap)
    fbuf = np.zeros(cap, vk.FEATURE_DTYPE)
    mbuf = np.zeros(cap, vk.MATCH_DTYPE)
    assert L.vksift_ext_pinHostMemory(fbuf.ctypes.data, fbuf.nbytes) == 0 and L.vksift_ext_pinHostMemory(mbuf.ctypes.data, mbuf.nbytes) == 0
    try:
        with vk.Instance(cfg, batch_capacity=n) as inst:
            ids = list(range(n))
            for order in ("pinned_first", "pageable_first"):
                inst.detectFeaturesBatch(imgs, 0)
                inst.matchFeaturesBatch(ids, ids[::-1])
                ref, got = {}, {}
                seq = ids if order == "pinned_first" else []
                if order == "pageable_first":
                    for i in ids[:3]:
                        ref[i] = inst.downloadFeatures(i)      # builds the staged copy: the pinned downloads below read it
                for i in ids:
                    cnt = inst.getFeaturesNumber(i)
                    fbuf[:] = 0
                    L.vksift_downloadFeatures(inst._h, fbuf.ctypes.data, i)
                    got[i] = fbuf[:cnt].copy()
                for i in ids:
                    if i not in ref:
                        ref[i] = inst.downloadFeatures(i)
                for i in ids:
                    assert len(got[i]) > 50 and got[i].tobytes() == ref[i].tobytes(), (order, i)
                for k in ids:
                    cnt = L.vksift_ext_getMatchesNumberBatch(inst._h, k)
                    mbuf[:] = 0
                    L.vksift_ext_downloadMatchesBatch(inst._h, k, mbuf.ctypes.data)
                    assert mbuf[:cnt].tobytes() == inst.downloadMatchesBatch(k).tobytes(), (order, k)
    finally:
        assert L.vksift_ext_unpinHostMemory(fbuf.ctypes.data) == 0 and L.vksift_ext_unpinHostMemory(mbuf.ctypes.data) == 0
