"""Parity of the path bench.py times — not of a path that resembles it.

bench.py's step is: a batch_capacity = 512 instance (two pyramid buffers, overlapped detections),
vksift_ext_detectFeaturesBatchDevice on 512 device-resident 640x480 frames, vksift_ext_matchFeaturesBatch(ids, ids) of all 512
self-pairs (launch sequences of 64 inside), the next step queued right behind it WITHOUT a host synchronisation. The reference makes a new detection wait
for the running pipelines (src/vulkansift/vulkansift.c:326-327, include/vulkansift/vulkansift.h:43-47); here the ordering is done
with events between streams (vksift_detect.c: ev_pyr_free, ev_desc_start, ev_input_free), which is exactly what these tests load:

* three (or two) steps back to back, each on a DIFFERENT frame set, so a buffer recycled too early or a stale pyramid shows;
* after the last step: every buffer byte-equal to the plain single-image vksift_detectFeatures on the same frame, sampled
  frames byte-equal to the oracle, sampled self-match / pair-match records byte-equal to the oracle;
* with and without profiling (the timed region of bench.py runs with it on: the event-set recycling is part of the path).

The C5-share shape (64 x 1080p, up-sampling on, consecutive pairs in both directions, src/examples/test_sift_match.cpp:67-80)
gets the same treatment.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


_FRAMES = {}


def _frames(vk, n, w, h, seed0):
    key = (n, w, h, seed0)
    if key not in _FRAMES:                       # 85 ms per 640x480 frame on the host: generated once per module
        _FRAMES[key] = np.stack([vk.gen_synthetic_image(seed0 + i, w, h) for i in range(n)])
    return _FRAMES[key]


def _feat_sets(vk, n, w, h, seed0):
    """three frame sets for three steps: only the last one is a set of fresh synthetic frames; the earlier ones are flips of
    it (different content in every buffer at no generation cost)"""
    last = _frames(vk, n, w, h, seed0)
    return [np.ascontiguousarray(last[::-1, ::-1, :]), np.ascontiguousarray(last[:, :, ::-1]), last]


def _single_image_reference(vk, frames, **cfg_kw):
    out = []
    with vk.Instance(vk.default_config(**cfg_kw)) as inst:
        for f in frames:
            inst.detectFeatures(f, 0)
            out.append(inst.downloadFeatures(0))
    return out


def _bench_frames(vk, B, W, H):
    """bench.py's frame set: 128 generated frames and their three mirror images"""
    gen = _frames(vk, 128, W, H, 0x5EED0000)
    variants = [gen, gen[:, :, ::-1], gen[:, ::-1, :], gen[:, ::-1, ::-1]]
    return np.ascontiguousarray(np.concatenate([variants[k % 4] for k in range(B // 128)]))


@pytest.mark.parametrize("profiling,two_calls", [(True, False), (False, True)], ids=["profiled", "unprofiled_two_match_calls"])
def test_bench_step_back_to_back_equals_single_image_and_oracle(vk, oracle, profiling, two_calls):
    """bench.py's default step: 512 device-resident frames per detection call, one 512-pair self-matching (or two of 256)"""
    import torch

    B, W, H = 512, 640, 480
    last = _bench_frames(vk, B, W, H)
    sets = [np.roll(last, 2 * 128, axis=0), np.roll(last, 128, axis=0), last]     # every buffer sees three different frames
    d_sets = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in sets]
    torch.cuda.synchronize()
    calls = [list(range(0, 256)), list(range(256, 512))] if two_calls else [list(range(B))]
    cfg = vk.default_config(sift_buffer_count=B, input_image_max_size=W * H)
    with vk.Instance(cfg, batch_capacity=B) as inst:
        inst.setProfiling(profiling)
        for k in range(3):                       # bench.py's step(), three times, nothing in between
            inst.detectFeaturesBatchDevice(d_sets[k].data_ptr(), B, W, H, 0)
            for ids in calls:
                inst.matchFeaturesBatch(ids, ids)
        # the matches of the last call first (the accessors below synchronise)
        ids = calls[-1]
        picks = (0, 1, 63, 64, 200, len(ids) - 1)
        matches = {ids[k]: inst.downloadMatchesBatch(k) for k in picks}
        counts = [inst.getFeaturesNumber(i) for i in range(B)]
        feats = [inst.downloadFeatures(i) for i in range(B)]
        if profiling:
            acc = inst.getAccumulatedDetectTimings()
            assert acc["nb_calls"] == 3 and acc["pyramid_ms"] > 0 and acc["scan_ms"] > 0 and acc["total_ms"] > 0
    assert min(counts) > 1000
    single = _single_image_reference(vk, sets[2], input_image_max_size=W * H)
    for i in range(B):
        assert counts[i] == len(single[i]), i
        assert feats[i].tobytes() == single[i].tobytes(), i
    ocfg = oracle.default_config(math_mode=1)
    for i in (0, 17, 127, 128, 300, 383, 384, 511):
        ref, _ = oracle.detect(ocfg, sets[2][i])
        assert feats[i].tobytes() == ref.tobytes(), i
    for b, m in matches.items():
        assert m.tobytes() == oracle.match_2nn(feats[b], feats[b]).tobytes(), b


def test_bench_step_host_protocol_back_to_back(vk, oracle):
    """the same three steps through the host-image entry (vksift_ext_detectFeaturesBatch): the staging buffer and d_input are
    recycled by the next call while the previous detection may still be running"""
    B, W, H = 128, 640, 480
    sets = _feat_sets(vk, B, W, H, 0x5EED0000)
    cfg = vk.default_config(sift_buffer_count=B, input_image_max_size=W * H)
    with vk.Instance(cfg, batch_capacity=B) as inst:
        for k in range(3):
            inst.detectFeaturesBatch(list(sets[k]), 0)
            inst.matchFeaturesBatch(list(range(B)), list(range(B)))       # one call, 128 pairs (two runs of 64 inside)
        matches = [inst.downloadMatchesBatch(k) for k in range(B)]      # packed download from the second pair on
        feats = [inst.downloadFeatures(i) for i in range(B)]
    single = _single_image_reference(vk, sets[2], input_image_max_size=W * H)
    for i in range(B):
        assert feats[i].tobytes() == single[i].tobytes(), i
    for k in (0, 1, 63, 64, 100, 127):
        assert matches[k].tobytes() == oracle.match_2nn(feats[k], feats[k]).tobytes(), k
    for k in range(B):                                                   # every pair: shape + self-match property
        assert len(matches[k]) == len(feats[k]) and np.array_equal(matches[k]["idx_a"], np.arange(len(feats[k]), dtype=np.uint32))
        assert np.all(matches[k]["dist_a_b1"] == 0)


def test_filtered_matching_of_128_pairs_in_one_call(vk, oracle):
    """vksift_ext_matchFeaturesFiltered beyond one 64-slot launch sequence: 128 pairs (i, i+1 mod 128) with cross-check, the
    forward + reverse matchings and the filter run in two runs of 64 slots; sampled pairs of both runs against the CPU filter"""
    B, W, H = 128, 640, 480
    frames = _frames(vk, B, W, H, 0x5EED0000)
    cfg = vk.default_config(sift_buffer_count=B, input_image_max_size=W * H)
    a = list(range(B))
    b = [(i + 1) % B for i in a]
    with vk.Instance(cfg, batch_capacity=B) as inst:
        inst.detectFeaturesBatch(list(frames), 0)
        inst.matchFeaturesFiltered(a, b, 0.8, True)
        got = {k: inst.downloadFilteredMatches(k) for k in (0, 63, 64, 127)}
        fwd = {k: inst.downloadMatchesBatch(k) for k in (0, 63, 64, 127)}
        feats = {i: inst.downloadFeatures(i) for i in (0, 1, 63, 64, 65, 127)}
    for k in (0, 63, 64, 127):
        fa, fb = feats[a[k]], feats[b[k]]
        m12, m21 = oracle.match_2nn(fa, fb), oracle.match_2nn(fb, fa)
        assert fwd[k].tobytes() == m12.tobytes(), k
        ra, rb = oracle.filter_matches(m12, m21, 0.8, True)
        assert np.array_equal(got[k]["idx_a"], ra) and np.array_equal(got[k]["idx_b"], rb), k


def test_256_distinct_pairs_over_512_buffers_in_one_call(vk, oracle):
    """one launch sequence of VKSIFT_HIP_MATCH_SLOTS = 256 pairs whose two sides name 256 DISTINCT not-yet-cached buffers each
    (even against odd, 512 buffers): the matcher cache of every one of them has to be gathered before the kernels read it
    (vksift_match.c: refresh_match_cache walks the ids in passes of 128; round 4 stopped after the first 128 and matched slots
    128..255 against stale cache entries). Small frames so that the oracle checks EVERY pair."""
    B, W, H = 512, 128, 96
    frames = np.stack([vk.gen_synthetic_image(0xC0FFEE + i, W, H) for i in range(B)])
    cfg = vk.default_config(sift_buffer_count=B, input_image_max_size=W * H, max_nb_sift_per_buffer=4096)
    a = list(range(0, B, 2))
    b = list(range(1, B, 2))
    with vk.Instance(cfg, batch_capacity=B) as inst:
        inst.detectFeaturesBatch(list(frames), 0)
        inst.matchFeaturesBatch(a, b)                      # nothing cached yet on either side
        fwd = [inst.downloadMatchesBatch(k) for k in range(len(a))]
        inst.matchFeaturesBatch(b, a)                      # everything cached now: the reverse direction
        rev = [inst.downloadMatchesBatch(k) for k in range(len(a))]
        feats = [inst.downloadFeatures(i) for i in range(B)]
    assert min(len(f) for f in feats) >= 2
    for k in range(len(a)):
        assert fwd[k].tobytes() == oracle.match_2nn(feats[a[k]], feats[b[k]]).tobytes(), k
        assert rev[k].tobytes() == oracle.match_2nn(feats[b[k]], feats[a[k]]).tobytes(), k


def test_pipelined_two_buffer_sets_like_the_bench_leg(vk, oracle):
    """bench.py's value_host_input_pipelined: 2 x 128 buffers, detection of the next batch queued before the results of the current
    one are fetched. Accessors must wait for the detection that filled THEIR buffer (sequence-numbered completion events), results
    must be those of the right frame set, and vksift_isBufferAvailable must tell the two sets apart."""
    B, W, H = 128, 640, 480
    sets = _feat_sets(vk, B, W, H, 0x5EED0000)
    cfg = vk.default_config(sift_buffer_count=2 * B, input_image_max_size=W * H)
    ids = [list(range(B)), list(range(B, 2 * B))]
    got = {}
    with vk.Instance(cfg, batch_capacity=B) as inst:
        inst.detectFeaturesBatch(list(sets[0]), 0)
        inst.matchFeaturesBatch(ids[0], ids[0])
        for it in range(3):
            cur, nxt = it & 1, (it & 1) ^ 1
            if it + 1 < 3:
                inst.detectFeaturesBatch(list(sets[it + 1]), nxt * B)
            feats = [inst.downloadFeatures(i) for i in ids[cur]]
            matches = [inst.downloadMatchesBatch(k) for k in (0, 64, 127)]
            if it + 1 < 3:
                # the set just fetched is idle, the set being detected is not (unless the GPU has already finished it)
                assert inst.isBufferAvailable(ids[cur][5])
                inst.matchFeaturesBatch(ids[nxt], ids[nxt])
            got[it] = (feats, matches)
    for it in (1, 2):
        single = _single_image_reference(vk, sets[it][::17], input_image_max_size=W * H)
        feats, matches = got[it]
        for j, i in enumerate(range(0, B, 17)):
            assert feats[i].tobytes() == single[j].tobytes(), (it, i)
        for k, m in zip((0, 64, 127), matches):
            assert m.tobytes() == oracle.match_2nn(feats[k], feats[k]).tobytes(), (it, k)


def test_c5_share_device_input_back_to_back(vk, oracle):
    """one GPU's share of BASELINE config 5, as tools/bench_configs.py and bench.py's c5 leg run it: 64 x 1080p (up-sampling on)
    from device memory + the 32 consecutive pairs matched in both directions, twice back to back on different frames"""
    import torch

    B, W, H = 64, 1920, 1080
    last = _frames(vk, B, W, H, 0x5EED0000)
    sets = [np.ascontiguousarray(last[:, ::-1, :]), last]
    d_sets = [torch.from_numpy(s).cuda() for s in sets]
    torch.cuda.synchronize()
    even, odd = list(range(0, B, 2)), list(range(1, B, 2))
    cfg = vk.default_config(sift_buffer_count=B, input_image_max_size=W * H)
    with vk.Instance(cfg, batch_capacity=B) as inst:
        for k in range(2):
            inst.detectFeaturesBatchDevice(d_sets[k].data_ptr(), B, W, H, 0)
            inst.matchFeaturesBatch(even, odd)
            inst.matchFeaturesBatch(odd, even)
        rev = {(odd[k], even[k]): inst.downloadMatchesBatch(k) for k in (0, 13, 31)}
        feats = [inst.downloadFeatures(i) for i in range(B)]
        inst.matchFeaturesBatch(even, odd)
        fwd = {(even[k], odd[k]): inst.downloadMatchesBatch(k) for k in (0, 13, 31)}
    assert min(len(f) for f in feats) > 5000
    single = _single_image_reference(vk, last, input_image_max_size=W * H)
    for i in range(B):
        assert feats[i].tobytes() == single[i].tobytes(), i
    ocfg = oracle.default_config(math_mode=1)
    for i in (5, 58):
        ref, _ = oracle.detect(ocfg, last[i])
        assert feats[i].tobytes() == ref.tobytes(), i
    for (a, b), m in {**rev, **fwd}.items():
        assert m.tobytes() == oracle.match_2nn(feats[a], feats[b]).tobytes(), (a, b)


@pytest.mark.parametrize("seed", [7, 8])
def test_random_operation_sequences_on_a_pingpong_instance(vk, oracle, seed):
    """250 random API calls on an instance with batch_capacity = 8 (two pyramid buffers, overlapped detections, batched
    download): batched detections of 8 frames, single detections, uploads, single / batched / filtered matches whose records
    are fetched only later, accessors in any order. Every observable result must equal a model built from the oracle.
    Unlike tests/test_gpu_api_scenarios.py::test_random_operation_sequences_follow_the_model, most operations here do NOT
    synchronise, so runs of detections and matches queue up behind each other."""
    rng = np.random.default_rng(seed)
    W, H = 320, 240
    imgs = [vk.gen_synthetic_image(800 + i, W, H) for i in range(12)]
    other = vk.gen_synthetic_image(820, 256, 192)
    ocfg = oracle.default_config(math_mode=1)
    ref = [oracle.detect(ocfg, im)[0] for im in imgs]
    ref_other = oracle.detect(ocfg, other)[0]
    nbuf = 12
    model = {b: np.zeros(0, vk.FEATURE_DTYPE) for b in range(nbuf)}
    pending = None            # (kind, expected records per pair) of the last matching call, not fetched yet
    cfg = vk.default_config(sift_buffer_count=nbuf, input_image_max_size=W * H)
    ops = ["batch", "detect", "detect_other", "match", "batchmatch", "fetch", "download", "count", "upload", "filtered"]
    prob = [0.2, 0.1, 0.05, 0.15, 0.1, 0.12, 0.13, 0.05, 0.05, 0.05]
    with vk.Instance(cfg, batch_capacity=8) as inst:
        for step in range(250):
            op = rng.choice(ops, p=prob)
            if op == "batch":
                b = int(rng.integers(nbuf - 7))
                first = int(rng.integers(len(imgs) - 7))
                inst.detectFeaturesBatch(imgs[first:first + 8], b)
                for i in range(8):
                    model[b + i] = ref[first + i]
            elif op == "detect":
                k, b = int(rng.integers(len(imgs))), int(rng.integers(nbuf))
                inst.detectFeatures(imgs[k], b)
                model[b] = ref[k]
            elif op == "detect_other":            # a resolution change between overlapped detections
                b = int(rng.integers(nbuf))
                inst.detectFeatures(other, b)
                model[b] = ref_other
            elif op == "upload":
                k, b = int(rng.integers(len(imgs))), int(rng.integers(nbuf))
                n = int(rng.integers(2, len(ref[k]) + 1))
                inst.uploadFeatures(ref[k][:n].copy(), b)
                model[b] = ref[k][:n]
            elif op == "count":
                b = int(rng.integers(nbuf))
                assert inst.getFeaturesNumber(b) == len(model[b]), (step, op, b)
            elif op == "download":
                b = int(rng.integers(nbuf))
                assert inst.downloadFeatures(b).tobytes() == model[b].tobytes(), (step, op, b)
            elif op == "match":
                a, b = int(rng.integers(nbuf)), int(rng.integers(nbuf))
                if len(model[a]) == 0 or len(model[b]) < 2:
                    continue
                inst.matchFeatures(a, b)
                pending = ("single", [oracle.match_2nn(model[a], model[b])])
            elif op == "batchmatch":
                n = int(rng.integers(2, 9))
                pa = [int(x) for x in rng.integers(nbuf, size=n)]
                pb = [int(x) for x in rng.integers(nbuf, size=n)]
                if any(len(model[a]) == 0 for a in pa) or any(len(model[b]) < 2 for b in pb):
                    continue
                inst.matchFeaturesBatch(pa, pb)
                pending = ("batch", [oracle.match_2nn(model[a], model[b]) for a, b in zip(pa, pb)])
            elif op == "fetch":
                if pending is None:
                    continue
                kind, exp = pending
                if kind == "single":
                    assert inst.downloadMatches().tobytes() == exp[0].tobytes(), (step, op)
                else:
                    for k, e in enumerate(exp):
                        assert inst.downloadMatchesBatch(k).tobytes() == e.tobytes(), (step, op, k)
            else:
                a, b = int(rng.integers(nbuf)), int(rng.integers(nbuf))
                if len(model[a]) < 2 or len(model[b]) < 2:
                    continue
                inst.matchFeaturesFiltered([a], [b], 0.8, True)
                got = inst.downloadFilteredMatches(0)
                m12 = oracle.match_2nn(model[a], model[b])
                ra, rb = oracle.filter_matches(m12, oracle.match_2nn(model[b], model[a]), 0.8, True)
                assert np.array_equal(got["idx_a"], ra) and np.array_equal(got["idx_b"], rb), (step, op, a, b)
                pending = ("single", [m12])       # the forward records stay available as pair 0
        for b in range(nbuf):
            assert inst.downloadFeatures(b).tobytes() == model[b].tobytes(), ("final", b)
