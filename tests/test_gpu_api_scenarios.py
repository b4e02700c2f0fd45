"""GPU tests of the API call sequences the reference's example programs exercise, the
golden fixtures, and the batched / capacity edge cases. Everything goes through the C-ABI."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _eq_feats(a, b):
    assert len(a) == len(b)
    for name in a.dtype.names:
        x, y = a[name], b[name]
        if x.dtype.kind == "f":
            x, y = x.view(np.uint32), y.view(np.uint32)
        assert np.array_equal(x, y), name


def test_golden_features_and_matches(vk):
    img = np.load(os.path.join(G, "img_160x120.npy"))
    meta = json.load(open(os.path.join(G, "meta.json")))
    with vk.Instance(vk.default_config()) as inst:
        inst.detectFeatures(img, 0)
        assert inst.getFeaturesNumber(0) == meta["det"]["n"]
        _eq_feats(inst.downloadFeatures(0), np.load(os.path.join(G, "feats_160x120_default_det.npy")))
    cfg = vk.default_config(use_input_upsampling=False, descriptor_format=vk.VKSIFT_DESCRIPTOR_FORMAT_VLFEAT, max_nb_orientation_per_keypoint=0,
                            use_hardware_interpolated_blur=False)
    with vk.Instance(cfg) as inst:
        inst.detectFeatures(img, 1)
        _eq_feats(inst.downloadFeatures(1), np.load(os.path.join(G, "feats_160x120_noups_vlfeat_det.npy")))
        a = np.zeros(256, vk.FEATURE_DTYPE)
        b = np.zeros(300, vk.FEATURE_DTYPE)
        a["descriptor"] = np.load(os.path.join(G, "desc_a.npy"))
        b["descriptor"] = np.load(os.path.join(G, "desc_b.npy"))
        inst.uploadFeatures(a, 0)
        inst.uploadFeatures(b, 1)
        inst.matchFeatures(0, 1)
        m = inst.downloadMatches()
        ref = np.load(os.path.join(G, "matches_a_b.npy"))
        for name in ref.dtype.names:
            x, y = m[name], ref[name]
            if x.dtype.kind == "f":
                x, y = x.view(np.uint32), y.view(np.uint32)
            assert np.array_equal(x, y), name


def test_detect_match_sequence_like_test_sift_match(vk, oracle):
    """test_sift_match.cpp:67-85: detect(0), detect(1), match(0,1), download, match(1,0), download, download features"""
    img1 = vk.gen_synthetic_image(101, 320, 240)
    img2 = np.roll(img1, (3, 5), axis=(0, 1))
    with vk.Instance(vk.default_config()) as inst:
        inst.detectFeatures(img1, 0)
        inst.detectFeatures(img2, 1)
        inst.matchFeatures(0, 1)
        n01 = inst.getMatchesNumber()
        m01 = inst.downloadMatches()
        inst.matchFeatures(1, 0)
        m10 = inst.downloadMatches()
        f0, f1 = inst.downloadFeatures(0), inst.downloadFeatures(1)
    assert n01 == len(f0) == len(m01) and len(m10) == len(f1)
    r01, r10 = oracle.match_2nn(f0, f1), oracle.match_2nn(f1, f0)
    assert np.array_equal(m01["idx_b1"], r01["idx_b1"]) and np.array_equal(m01["idx_b2"], r01["idx_b2"])
    assert np.array_equal(m10["idx_b1"], r10["idx_b1"]) and np.array_equal(m10["idx_b2"], r10["idx_b2"])
    # the shifted copy must be recognisable: cross-check + Lowe ratio as in test_sift_match.cpp:90-107
    good = [(m["idx_a"], m["idx_b1"]) for m in m01 if m["dist_a_b1"] < 0.75 * m["dist_a_b2"] and m10[m["idx_b1"]]["idx_b1"] == m["idx_a"]]
    assert len(good) > 0.3 * len(f0)
    dx = np.array([f1[j]["x"] - f0[i]["x"] for i, j in good])
    dy = np.array([f1[j]["y"] - f0[i]["y"] for i, j in good])
    assert abs(np.median(dx) - 5) < 0.5 and abs(np.median(dy) - 3) < 0.5


def test_download_upload_roundtrip_like_test_sift_gpu_debug(vk):
    """test_sift_gpu_debug.cpp:92-124: detect -> download -> upload -> match"""
    img = vk.gen_synthetic_image(102, 256, 256)
    with vk.Instance(vk.default_config()) as inst:
        inst.detectFeatures(img, 0)
        f = inst.downloadFeatures(0)
        inst.uploadFeatures(f, 1)
        assert inst.getFeaturesNumber(1) == len(f)
        _eq_feats(inst.downloadFeatures(1), f)
        inst.matchFeatures(0, 1)
        m = inst.downloadMatches()
        assert inst.getFeaturesNumber(0) == len(f)  # buffer 0 is now "packed", same count
        _eq_feats(inst.downloadFeatures(0), f)
    uniq = np.unique(f["descriptor"], axis=0, return_index=True)[1]
    assert np.all(m["dist_a_b1"] == 0)
    assert np.array_equal(m["idx_b1"][uniq], m["idx_a"][uniq])


def test_error_callback_contract_like_test_sift_error_handling(vk):
    """test_sift_error_handling.cpp:52-60: invalid buffer index -> callback(VKSIFT_INVALID_INPUT_ERROR), instance stays usable"""
    cfg = vk.default_config(sift_buffer_count=3)
    img = vk.gen_synthetic_image(103, 128, 128)
    with vk.Instance(cfg) as inst:
        for idx in range(0, 6):
            if idx < 3:
                assert inst.getFeaturesNumber(idx) == 0
            else:
                with pytest.raises(vk.VksiftError) as e:
                    inst.getFeaturesNumber(idx)
                assert e.value.code == vk.VKSIFT_INVALID_INPUT_ERROR
        with pytest.raises(vk.VksiftError):
            inst.detectFeatures(img, 7)
        with pytest.raises(vk.VksiftError):
            inst.detectFeatures(np.zeros((16, 16), np.uint8), 0)          # < 1024 pixels
        with pytest.raises(vk.VksiftError):
            inst.detectFeatures(np.zeros((2000, 2000), np.uint8), 0)      # > input_image_max_size
        with pytest.raises(vk.VksiftError):
            inst.matchFeatures(0, 3)
        with pytest.raises(vk.VksiftError):
            inst.downloadScaleSpaceImage(0, 6)
        with pytest.raises(vk.VksiftError):
            inst.downloadDoGImage(9, 0)
        inst.detectFeatures(img, 2)                                        # still usable
        assert inst.getFeaturesNumber(2) > 0
        inst.presentDebugFrame()                                           # warning + no-op


def test_invalid_config_rejected(vk):
    import ctypes as C
    for kw in ({"input_image_max_size": 100}, {"sift_buffer_count": 0}, {"max_nb_sift_per_buffer": 0}, {"nb_scales_per_octave": 0},
               {"seed_scale_sigma": 0.5}, {"intensity_threshold": -1.0}, {"pyramid_precision_mode": 7}):
        cfg = vk.default_config(**kw)
        h = C.c_void_p(None)
        vk.load()
        assert vk.lib().vksift_createInstance(C.byref(h), C.byref(cfg)) == vk.VKSIFT_INVALID_INPUT_ERROR and not h


def test_async_contract_and_buffer_availability(vk):
    img = vk.gen_synthetic_image(104, 640, 480)
    with vk.Instance(vk.default_config()) as inst:
        assert inst.isBufferAvailable(0) and inst.isBufferAvailable(1)
        inst.detectFeatures(img, 0)
        assert inst.isBufferAvailable(1)          # not the target of the running pipeline
        n = inst.getFeaturesNumber(0)             # blocks
        assert inst.isBufferAvailable(0) and n > 100
        inst.detectFeatures(img, 1)
        inst.matchFeatures(0, 1)
        inst.downloadMatches()
        assert inst.isBufferAvailable(0) and inst.isBufferAvailable(1)


def test_resolution_change_and_scale_space_queries(vk, oracle):
    """test_sift_show_pyr.cpp:45-76 + a resolution switch between detections"""
    with vk.Instance(vk.default_config()) as inst:
        for (w, h) in ((320, 200), (131, 257), (320, 200)):
            img = vk.gen_synthetic_image(105, w, h)
            inst.detectFeatures(img, 0)
            pyr = oracle.Pyramid(oracle.default_config(math_mode=1), img)
            assert inst.getScaleSpaceNbOctaves() == pyr.nb_octaves
            o = pyr.nb_octaves - 1
            assert inst.getScaleSpaceOctaveResolution(o) == pyr.resolution(o)
            assert np.array_equal(inst.downloadScaleSpaceImage(o, 5), pyr.gauss(o, 5))
            assert np.array_equal(inst.downloadDoGImage(o, 4), pyr.dog(o, 4))
            ref, _ = pyr.detect()
            _eq_feats(inst.downloadFeatures(0), ref)


def test_batch_detect_equals_single(vk):
    imgs = [vk.gen_synthetic_image(200 + i, 300, 220) for i in range(5)]
    cfg = vk.default_config(sift_buffer_count=6)
    with vk.Instance(cfg, batch_capacity=5) as inst:
        inst.detectFeaturesBatch(imgs, 1)
        batch = [inst.downloadFeatures(1 + i) for i in range(5)]
        singles = []
        for i, im in enumerate(imgs):
            inst.detectFeatures(im, 0)
            singles.append(inst.downloadFeatures(0))
    for b, s in zip(batch, singles):
        assert len(s) > 50
        _eq_feats(b, s)


def test_batched_download_equals_per_buffer_download(vk, oracle):
    """detections of 8 images and more are downloaded through one packed copy of all their buffers (vksift_hip_pack_features): same
    records as the per-section copies, also with clamped sections, after an upload into one buffer of the range, after a
    single-image detection into another, and in any download order"""
    imgs = [vk.gen_synthetic_image(900 + i, 224 + 16 * (i % 2), 160) for i in range(11)]
    imgs = [im[:, :224].copy() for im in imgs]                     # one resolution per batched call
    ocfg = oracle.default_config(math_mode=1, max_nb_sift_per_buffer=260)
    refs = [oracle.detect(ocfg, im)[0] for im in imgs]
    assert max(len(r) for r in refs) <= 260 and any(sum(oracle.detect(ocfg, im)[1]) > len(r) for im, r in zip(imgs, refs))   # some sections clamp
    vk.lib().vksift_setLogLevel(vk.VKSIFT_NO_LOG)
    try:
        with vk.Instance(vk.default_config(max_nb_sift_per_buffer=260, sift_buffer_count=13), batch_capacity=11) as inst:
            inst.detectFeaturesBatch(imgs, 2)
            for i in (10, 0, 5, 3, 9, 1, 2, 4, 6, 7, 8):              # served from the packed copy
                _eq_feats(inst.downloadFeatures(2 + i), refs[i])
            up = refs[4][:17].copy()
            inst.uploadFeatures(up, 2 + 6)                           # buffer 8 now holds uploaded features
            _eq_feats(inst.downloadFeatures(8), up)
            _eq_feats(inst.downloadFeatures(7), refs[5])             # its neighbours still the detection's
            inst.detectFeatures(imgs[0], 4)                          # a single detection into the range
            _eq_feats(inst.downloadFeatures(4), refs[0])
            _eq_feats(inst.downloadFeatures(5), refs[3])
            inst.detectFeaturesBatch(imgs[:8], 0)                    # a new batch over part of the range
            for i in range(8):
                _eq_feats(inst.downloadFeatures(i), refs[i])
            _eq_feats(inst.downloadFeatures(12), refs[10])
    finally:
        vk.lib().vksift_setLogLevel(vk.VKSIFT_LOG_INFO)


def test_section_overflow_counts_and_clamps(vk, oracle):
    """max_nb_sift_per_buffer too small: counters keep counting, stores are dropped (ExtractKeypoints.comp:208-212)"""
    img = vk.gen_synthetic_image(106, 320, 240)
    vcfg = vk.default_config(max_nb_sift_per_buffer=100)
    ocfg = oracle.default_config(math_mode=1, max_nb_sift_per_buffer=100)
    vk.lib().vksift_setLogLevel(vk.VKSIFT_NO_LOG)
    try:
        with vk.Instance(vcfg) as inst:
            inst.detectFeatures(img, 0)
            got = inst.downloadFeatures(0)
    finally:
        vk.lib().vksift_setLogLevel(vk.VKSIFT_LOG_INFO)
    ref, counts = oracle.detect(ocfg, img)
    assert sum(counts) > 100 and len(ref) <= 100
    _eq_feats(got, ref)


def test_nb_octaves_config_and_more_scales(vk, oracle):
    img = vk.gen_synthetic_image(107, 256, 192)
    for kw in ({"nb_octaves": 2}, {"nb_scales_per_octave": 4}, {"seed_scale_sigma": 2.0}):
        vcfg = vk.default_config(**kw)
        ocfg = oracle.default_config(math_mode=1, **kw)
        with vk.Instance(vcfg) as inst:
            inst.detectFeatures(img, 0)
            got = inst.downloadFeatures(0)
        ref, _ = oracle.detect(ocfg, img)
        assert len(ref) > 10
        _eq_feats(got, ref)


def test_libm_oracle_within_tolerance(vk, oracle):
    """HIP path vs the oracle with independent libm math: the stated float tolerance of the parity claim."""
    img = vk.gen_synthetic_image(108, 640, 480)
    with vk.Instance(vk.default_config()) as inst:
        inst.detectFeatures(img, 0)
        got = inst.downloadFeatures(0)
    ref, _ = oracle.detect(oracle.default_config(math_mode=0), img)
    assert abs(len(got) - len(ref)) <= max(2, len(ref) // 200)
    # set-based comparison. Positions involve no transcendental function, so they are bit-identical between the two
    # math back-ends and can key the match; the orientation half-bin separates multi-orientation copies.
    def key(f):
        return (int(f["octave_idx"]), int(f["scale_idx"]), float(f["scale_x"]), float(f["scale_y"]),
                int(round(float(f["orientation"]) * 36 / (2 * np.pi) * 2)))
    rmap = {key(f): f for f in ref}
    hit = [(g, rmap[key(g)]) for g in got if key(g) in rmap]
    assert len(hit) >= 0.995 * len(ref)
    dx = np.array([abs(g["x"] - r["x"]) + abs(g["y"] - r["y"]) for g, r in hit])
    ds = np.array([abs(g["sigma"] / r["sigma"] - 1) for g, r in hit])
    do = np.array([abs(g["orientation"] - r["orientation"]) for g, r in hit])
    assert dx.max() < 1e-4 and ds.max() < 1e-5 and do.max() < 1e-4
    rms = np.array([np.sqrt(((g["descriptor"].astype(float) - r["descriptor"].astype(float)) ** 2).mean()) / 512.0 for g, r in hit])
    assert np.median(rms) < 1e-4 and np.percentile(rms, 99) < 1e-3


def test_batched_match_equals_single_calls(vk, oracle):
    """vksift_ext_matchFeaturesBatch: detection buffers (device-side counts, shared section layout) and a mixed
    case with uploaded buffers of different sizes (falls back to pair-by-pair launches)."""
    imgs = [vk.gen_synthetic_image(300 + i, 320, 240) for i in range(4)]
    cfg = vk.default_config(sift_buffer_count=8)
    with vk.Instance(cfg, batch_capacity=4) as inst:
        inst.detectFeaturesBatch(imgs, 0)
        inst.matchFeaturesBatch([0, 1, 2, 3], [1, 2, 3, 0])     # enqueued while the detection is still running
        got = [inst.downloadMatchesBatch(i) for i in range(4)]
        feats = [inst.downloadFeatures(i) for i in range(4)]
        for i in range(4):
            ref = oracle.match_2nn(feats[i], feats[(i + 1) % 4])
            assert inst.getMatchesNumberBatch(i) == len(feats[i])
            for name in ("idx_a", "idx_b1", "idx_b2"):
                assert np.array_equal(got[i][name], ref[name]), (i, name)
            assert np.array_equal(got[i]["dist_a_b1"].view(np.uint32), ref["dist_a_b1"].view(np.uint32))
        # pair 0 is what the classic accessors see
        assert inst.getMatchesNumber() == len(feats[0])
        assert np.array_equal(inst.downloadMatches()["idx_b1"], got[0]["idx_b1"])
        # non-uniform layouts: uploaded buffers of different sizes
        inst.uploadFeatures(feats[0][:50], 4)
        inst.uploadFeatures(feats[1][:77], 5)
        inst.matchFeaturesBatch([4, 5, 0], [5, 4, 4])
        for i, (a, b) in enumerate([(feats[0][:50], feats[1][:77]), (feats[1][:77], feats[0][:50]), (feats[0], feats[0][:50])]):
            ref = oracle.match_2nn(a, b)
            m = inst.downloadMatchesBatch(i)
            assert np.array_equal(m["idx_b1"], ref["idx_b1"]) and np.array_equal(m["idx_b2"], ref["idx_b2"]), i
        with pytest.raises(vk.VksiftError):
            inst.matchFeaturesBatch([0] * 5, [1] * 5)             # more pairs than the batch capacity


def _np_filter(m12, m21, ratio, cross):
    """independent numpy statement of the CPU loop of the reference's examples (test_sift_match.cpp:90-107)"""
    keep = []
    for i in range(len(m12)):
        j = int(m12["idx_b1"][i])
        if cross and not (j < len(m21) and int(m21["idx_b1"][j]) == i):
            continue
        if not (np.float32(m12["dist_a_b1"][i]) / np.float32(m12["dist_a_b2"][i]) < np.float32(ratio)):
            continue
        if cross and not (np.float32(m21["dist_a_b1"][j]) / np.float32(m21["dist_a_b2"][j]) < np.float32(ratio)):
            continue
        keep.append((int(m12["idx_a"][i]), j))
    return np.array(keep, dtype=np.uint32).reshape(-1, 2)


@pytest.mark.parametrize("cross,ratio", [(True, 0.75), (False, 0.75), (True, 0.9)])
def test_filtered_matching_equals_cpu_filter(vk, oracle, cross, ratio):
    """vksift_ext_matchFeaturesFiltered == 2x vksift_matchFeatures + the reference examples' CPU filter"""
    rng = np.random.default_rng(5)
    a = vk.gen_synthetic_descriptors(71, 1500)
    b = vk.gen_synthetic_descriptors(72, 1300)
    # plant true correspondences (noisy copies) so that the ratio test has survivors, plus exact duplicates (zero distances)
    idx = rng.permutation(1300)[:600]
    noisy = a[:600].astype(np.int32) + rng.integers(-6, 7, (600, 128))
    b[idx] = np.clip(noisy, 0, 255).astype(np.uint8)
    b[idx[:5]] = a[:5]
    b[7] = b[3]
    fa = np.zeros(len(a), vk.FEATURE_DTYPE)
    fb = np.zeros(len(b), vk.FEATURE_DTYPE)
    fa["descriptor"] = a
    fb["descriptor"] = b
    with vk.Instance(vk.default_config()) as inst:
        inst.uploadFeatures(fa, 0)
        inst.uploadFeatures(fb, 1)
        inst.matchFeaturesFiltered([0], [1], ratio, cross)
        got = inst.downloadFilteredMatches(0)
        fwd = inst.downloadMatches()              # the forward 2-NN records stay available
        inst.matchFeatures(1, 0)
        rev = inst.downloadMatches()
    m12 = oracle.match_2nn(a, b)
    m21 = oracle.match_2nn(b, a)
    for name in ("idx_a", "idx_b1", "idx_b2"):
        assert np.array_equal(fwd[name], m12[name]) and np.array_equal(rev[name], m21[name])
    ra, rb = oracle.filter_matches(m12, m21, ratio, cross)
    ref2 = _np_filter(m12, m21, ratio, cross)
    assert np.array_equal(np.stack([ra, rb], 1).reshape(-1, 2), ref2)      # oracle == independent restatement
    assert len(got) == len(ra) and len(ra) > 100
    assert np.array_equal(got["idx_a"], ra) and np.array_equal(got["idx_b"], rb)
    assert np.array_equal(got["dist_a_b1"].view(np.uint32), m12["dist_a_b1"][ra].view(np.uint32))
    assert np.array_equal(got["dist_a_b2"].view(np.uint32), m12["dist_a_b2"][ra].view(np.uint32))


def test_filtered_matching_batch_after_detection(vk, oracle):
    """batched filtered matching straight after a batched detection (device-side counts), frame i against frame i+1"""
    imgs = [vk.gen_synthetic_image(200 + i, 320, 240) for i in range(3)]
    cfg = vk.default_config(sift_buffer_count=3)
    with vk.Instance(cfg, batch_capacity=3) as inst:
        inst.detectFeaturesBatch(imgs, 0)
        inst.matchFeaturesFiltered([0, 1], [1, 2], 0.8, True)
        got = [inst.downloadFilteredMatches(p) for p in range(2)]
        feats = [inst.downloadFeatures(i) for i in range(3)]
    for p, (ia, ib) in enumerate([(0, 1), (1, 2)]):
        m12 = oracle.match_2nn(feats[ia], feats[ib])
        m21 = oracle.match_2nn(feats[ib], feats[ia])
        ra, rb = oracle.filter_matches(m12, m21, 0.8, True)
        assert np.array_equal(got[p]["idx_a"], ra) and np.array_equal(got[p]["idx_b"], rb)


def test_graph_replay_gives_identical_results(vk, monkeypatch):
    """VKSIFT_GRAPH=1: the captured launch sequence is replayed for the second and third detection of the same shape"""
    imgs = [vk.gen_synthetic_image(300 + i, 320, 240) for i in range(3)]
    ref = []
    with vk.Instance(vk.default_config()) as inst:
        for im in imgs:
            inst.detectFeatures(im, 0)
            ref.append(inst.downloadFeatures(0))
    monkeypatch.setenv("VKSIFT_GRAPH", "1")
    with vk.Instance(vk.default_config()) as inst:
        for im, r in zip(imgs, ref):
            inst.detectFeatures(im, 0)            # 1st call captures, 2nd and 3rd replay (same buffer, same staging pointer)
            f = inst.downloadFeatures(0)
            assert f.tobytes() == r.tobytes()
        inst.detectFeatures(imgs[0], 1)           # another buffer: a second graph
        assert inst.downloadFeatures(1).tobytes() == ref[0].tobytes()
        top = inst.downloadScaleSpaceImage(0, 5)
        assert np.isfinite(top).all() and top.std() > 0


_SWITCH_PROBE = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from vulkansift_amd import api as vk
vk.lib().vksift_setLogLevel(vk.VKSIFT_LOG_ERROR)
import os
for kv in os.environ.get("PROBE_TUNE", "").split(","):      # "knob=value,...": the test knobs of vksift_hip_tune (include/vksift_hip.h)
    if kv:
        vk.lib().vksift_hip_tune(int(kv.split("=")[0]), int(kv.split("=")[1]))
imgs = [vk.gen_synthetic_image(400 + i, 352, 264) for i in range(2)]
h = hashlib.sha256()
with vk.Instance(vk.default_config(sift_buffer_count=2), batch_capacity=2) as inst:
    for rep in range(2):                      # second round: replayed graphs, recycled ping-pong buffers
        inst.detectFeaturesBatch(imgs, 0)
        for b in range(2):
            h.update(inst.downloadFeatures(b).tobytes())
        inst.matchFeatures(0, 1)
        h.update(inst.downloadMatches().tobytes())
    for s in range(6):
        h.update(inst.downloadScaleSpaceImage(1, s).tobytes())
imgs8 = [vk.gen_synthetic_image(500 + i, 208, 160) for i in range(8)]
with vk.Instance(vk.default_config(sift_buffer_count=8), batch_capacity=8) as inst:   # batches of 8 and more size their grids differently
    inst.detectFeaturesBatch(imgs8, 0)
    for b in range(8):
        h.update(inst.downloadFeatures(b).tobytes())
print("DIGEST", h.hexdigest())
"""


def test_every_runtime_switch_is_bit_identical(vk):
    """each optional code path in a fresh process: features, matches and planes must not change by a bit"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # PROBE_TUNE: knob 1 = wide-blur mask (0: two texels per lane everywhere), 2 = octaves per multi-octave launch, 3 = pointer form of the refinement,
    # 5 = lane width of the two-scale blur launch, 6 = the next scale-space behind (not beside) the matching queued before it,
    # 7 = the matcher's dense rows by the gather pass instead of the descriptor launch, 9 = rows per wave of the extrema scan,
    # 10 = scales S+1, S+2 one launch per octave instead of one per scale, 11 = a single host image copied to device memory first,
    # 12 = output rows per wave of the smallest strip-march launches
    variants = [{}, {"VKSIFT_BLUR_KERNEL": "tile"}, {"VKSIFT_PYR_PINGPONG": "1"}, {"VKSIFT_PYR_PINGPONG": "0"}, {"VKSIFT_PYR_PINGPONG": "2"},
                {"VKSIFT_GRAPH": "1"}, {"VKSIFT_GRAPH": "0"}, {"PROBE_TUNE": "2=2"}, {"PROBE_TUNE": "2=1"}, {"PROBE_TUNE": "3=1"}, {"PROBE_TUNE": "1=0"},
                {"PROBE_TUNE": "1=1048575"}, {"VKSIFT_BLUR_PAIR": "0"}, {"VKSIFT_FORK_SCALES": "0"}, {"VKSIFT_LDS_CHAIN": "0"}, {"VKSIFT_POST_FEATURES": "0"},
                {"VKSIFT_PYR_PLACEMENT": "0"}, {"VKSIFT_MATCH_PK": "0"}, {"PROBE_TUNE": "5=1"}, {"PROBE_TUNE": "5=2"}, {"PROBE_TUNE": "7=1"}, {"PROBE_TUNE": "9=32"}, {"PROBE_TUNE": "10=1"}, {"PROBE_TUNE": "11=1"}, {"PROBE_TUNE": "12=16"}, {"VKSIFT_PYR_PINGPONG": "1", "PROBE_TUNE": "6=1"}, {"VKSIFT_PYR_PINGPONG": "1", "PROBE_TUNE": "2=2,3=1", "VKSIFT_BLUR_PAIR": "0"}]
    digests = {}
    for v in variants:
        env = dict(os.environ)
        env.update(v)
        r = subprocess.run([sys.executable, "-c", _SWITCH_PROBE % root], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, (v, r.stdout[-2000:], r.stderr[-2000:])
        digests[str(v)] = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1]
    assert len(set(digests.values())) == 1, digests


def _sequence_seeds():
    import os
    extra = os.environ.get("VKSIFT_TEST_SEQUENCE_SEEDS")       # e.g. "7,8,9": more sequences for a longer soak
    return [99, 100, 101] + ([int(x) for x in extra.split(",")] if extra else [])


@pytest.mark.parametrize("seed", _sequence_seeds())
def test_random_operation_sequences_follow_the_model(vk, oracle, seed):
    """3 x 250 random API calls on one instance (detections of changing resolution into changing buffers, batched detections,
    uploads, matches, filtered matches, accessors in any order): every observable result must equal a trivial model built
    from the oracle — stresses the host bookkeeping (pending events, section tables, graph cache, staging reuse)"""
    rng = np.random.default_rng(seed)
    sizes = [(320, 240), (320, 240), (256, 192), (400, 300), (200, 160)]
    imgs = [vk.gen_synthetic_image(700 + i, w, h) for i, (w, h) in enumerate(sizes)]
    ocfg = oracle.default_config(math_mode=1)
    ref = [oracle.detect(ocfg, im)[0] for im in imgs]
    nbuf = 4
    model = {b: np.zeros(0, vk.FEATURE_DTYPE) for b in range(nbuf)}
    last_match = None
    cfg = vk.default_config(sift_buffer_count=nbuf, input_image_max_size=400 * 300)
    with vk.Instance(cfg, batch_capacity=2) as inst:
        for step in range(250):
            op = rng.choice(["detect", "batch", "upload", "count", "download", "match", "filtered", "avail"], p=[0.25, 0.1, 0.1, 0.1, 0.15, 0.15, 0.1, 0.05])
            if op == "detect":
                k, b = int(rng.integers(len(imgs))), int(rng.integers(nbuf))
                inst.detectFeatures(imgs[k], b)
                model[b] = ref[k]
            elif op == "batch":
                b = int(rng.integers(nbuf - 1))
                inst.detectFeaturesBatch([imgs[0], imgs[1]], b)
                model[b], model[b + 1] = ref[0], ref[1]
            elif op == "upload":
                k, b = int(rng.integers(len(imgs))), int(rng.integers(nbuf))
                n = int(rng.integers(0, len(ref[k]) + 1))
                inst.uploadFeatures(ref[k][:n].copy(), b)
                model[b] = ref[k][:n]
            elif op == "count":
                b = int(rng.integers(nbuf))
                assert inst.getFeaturesNumber(b) == len(model[b]), (step, op, b)
            elif op == "download":
                b = int(rng.integers(nbuf))
                assert inst.downloadFeatures(b).tobytes() == model[b].tobytes(), (step, op, b)
            elif op in ("match", "filtered"):
                a, b = int(rng.integers(nbuf)), int(rng.integers(nbuf))
                if len(model[a]) == 0 or len(model[b]) < 2:
                    continue
                m12 = oracle.match_2nn(model[a], model[b])
                if op == "match":
                    inst.matchFeatures(a, b)
                    got = inst.downloadMatches()
                    assert got.tobytes() == m12.tobytes(), (step, op, a, b)
                elif len(model[a]) >= 2:
                    inst.matchFeaturesFiltered([a], [b], 0.8, True)
                    got = inst.downloadFilteredMatches(0)
                    ra, rb = oracle.filter_matches(m12, oracle.match_2nn(model[b], model[a]), 0.8, True)
                    assert np.array_equal(got["idx_a"], ra) and np.array_equal(got["idx_b"], rb), (step, op, a, b)
            else:
                b = int(rng.integers(nbuf))
                inst.getFeaturesNumber(b)           # waits for anything pending on b
                assert inst.isBufferAvailable(b)
