"""Deferred submission of vksift_detectFeatures (vksift_internal.h: defer_enabled; include/vksift_ext.h: vksift_ext_getDeferredStats).

A caller of the reference's plain API (one image per call, vulkansift.c:315-344) who issues several detect calls in a row gets them
launched as one batched detection. Nothing a caller can observe may change: every buffer must hold, byte for byte, what the same
calls deliver with VKSIFT_DEFER=0 (every call launched at once) — whatever the order of buffers, resolutions, accessors and matchings —
and the calling patterns that never batch (detect + read, the steady state of a two-buffer ping-pong) must not be deferred at all."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _instance(vk, monkeypatch, defer, max_px, nbuf, batch_capacity=1, defer_max=None, **kw):
    monkeypatch.setenv("VKSIFT_DEFER", "1" if defer else "0")
    if defer_max is None:
        monkeypatch.delenv("VKSIFT_DEFER_MAX", raising=False)
    else:
        monkeypatch.setenv("VKSIFT_DEFER_MAX", str(defer_max))
    return vk.Instance(vk.default_config(input_image_max_size=max_px, sift_buffer_count=nbuf, **kw), batch_capacity=batch_capacity)


def _images(vk, n, w, h, seed):
    return [vk.gen_synthetic_image_family(seed + i, w, h, i % 3) for i in range(n)]


@pytest.mark.parametrize("w,h,n", [(320, 240, 24), (640, 480, 70), (97, 61, 9)])
def test_run_of_detect_calls_equals_immediate_launches(vk, monkeypatch, w, h, n):
    imgs = _images(vk, n, w, h, 9100 + w)
    with _instance(vk, monkeypatch, False, w * h, n) as inst:
        for i, img in enumerate(imgs):
            inst.detectFeatures(img, i)
        ref = [inst.downloadFeatures(i) for i in range(n)]
        assert inst.getDeferredStats() == (0, 0)
    assert sum(len(r) for r in ref) > 0
    with _instance(vk, monkeypatch, True, w * h, n) as inst:
        for rep in range(4):  # the capacity doubles batch by batch: the later runs go in fewer, larger launches
            for i, img in enumerate(imgs):
                inst.detectFeatures(img, i)
            order = range(n) if rep % 2 == 0 else reversed(range(n))
            for i in order:
                assert inst.getFeaturesNumber(i) == len(ref[i])
                assert inst.downloadFeatures(i).tobytes() == ref[i].tobytes(), (rep, i)
        batches, images = inst.getDeferredStats()
        # the first call of the first run goes at once; everything else was staged
        assert images == 4 * n - 1
        # the last run is one launch (n <= 128) or ceil(n / 128)
        inst.detectFeatures(imgs[0], 0)
        inst.detectFeatures(imgs[1], 1)
        b0 = inst.getDeferredStats()[0]
        for i, img in enumerate(imgs):
            inst.detectFeatures(img, i)
        assert inst.isBufferAvailable(n - 1) in (True, False)
        assert inst.getDeferredStats()[0] - b0 <= 2 + (n + 127) // 128


def test_single_detections_and_ping_pong_are_never_deferred(vk, monkeypatch):
    w, h = 320, 240
    imgs = _images(vk, 6, w, h, 555)
    with _instance(vk, monkeypatch, True, w * h, 2) as inst:
        ref = []
        for img in imgs:  # detect + read: the reference's own loop (vulkansift_wrapper.cpp:30-33)
            inst.detectFeatures(img, 0)
            ref.append(inst.downloadFeatures(0))
        assert inst.getDeferredStats() == (0, 0)
        # two-buffer ping-pong: the detection of frame k + 1 is queued before frame k is read. Its first two calls are a run of two
        # (staged, and the run after a run of two is staged from its first call on); from then on every run holds one call
        inst.detectFeatures(imgs[0], 0)
        for k in range(len(imgs)):
            if k + 1 < len(imgs):
                inst.detectFeatures(imgs[k + 1], (k + 1) & 1)
            assert inst.downloadFeatures(k & 1).tobytes() == ref[k].tobytes()
            if k == 1:
                steady = inst.getDeferredStats()
        assert steady == (2, 2) and inst.getDeferredStats() == steady


def test_mixed_orders_resolutions_and_matchings(vk, monkeypatch):
    """buffers out of order, named twice, two resolutions in one run, a narrow image that outgrows the reservation, matchings and
    uploads in between: the deferred instance against the immediate one, call by call."""
    rng = np.random.default_rng(77)
    shapes = [(320, 240), (200, 152), (139, 356), (320, 240)]
    pool = [vk.gen_synthetic_image_family(3000 + k, *shapes[k % 4], k % 3) for k in range(12)]
    nbuf = 8
    script = []
    for step in range(120):
        r = rng.integers(0, 100)
        if r < 60:
            script.append(("detect", int(rng.integers(0, len(pool))), int(rng.integers(0, nbuf))))
        elif r < 70:
            b = int(rng.integers(0, nbuf - 3))
            k = int(rng.integers(0, len(pool)))
            for j in range(3):  # a consecutive run of one resolution
                script.append(("detect", k, b + j))
        elif r < 80:
            script.append(("read", int(rng.integers(0, nbuf))))
        elif r < 88:
            script.append(("match", int(rng.integers(0, nbuf)), int(rng.integers(0, nbuf))))
        elif r < 93:
            script.append(("avail", int(rng.integers(0, nbuf))))
        elif r < 97:
            script.append(("count", int(rng.integers(0, nbuf))))
        else:
            script.append(("octaves",))

    def play(inst):
        out = []
        for op in script:
            if op[0] == "detect":
                inst.detectFeatures(pool[op[1]], op[2])
            elif op[0] == "read":
                out.append(inst.downloadFeatures(op[1]).tobytes())
            elif op[0] == "match":
                inst.matchFeatures(op[1], op[2])
                out.append(inst.downloadMatches().tobytes())
            elif op[0] == "avail":
                inst.isBufferAvailable(op[1])
            elif op[0] == "count":
                out.append(inst.getFeaturesNumber(op[1]))
            else:
                out.append(inst.getScaleSpaceNbOctaves())
        for b in range(nbuf):
            out.append(inst.downloadFeatures(b).tobytes())
        return out

    with _instance(vk, monkeypatch, False, 320 * 240, nbuf) as inst:
        ref = play(inst)
    with _instance(vk, monkeypatch, True, 320 * 240, nbuf) as inst:
        got = play(inst)
        assert inst.getDeferredStats()[1] > 10
    assert len(ref) == len(got)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert a == b, i


def test_invalid_arguments_are_reported_by_the_call_itself(vk, monkeypatch):
    w, h = 160, 120
    img = vk.gen_synthetic_image(5, w, h)
    with _instance(vk, monkeypatch, True, w * h, 4) as inst:
        inst.detectFeatures(img, 0)
        inst.detectFeatures(img, 1)  # staged
        with pytest.raises(vk.VksiftError):
            inst.detectFeatures(img, 4)  # no such buffer: reported now, not by the accessor that launches the batch
        with pytest.raises(vk.VksiftError):
            inst.detectFeatures(np.zeros((400, 400), np.uint8), 2)  # larger than the instance takes
        inst.detectFeatures(img, 2)
        a, b, c = inst.downloadFeatures(0), inst.downloadFeatures(1), inst.downloadFeatures(2)
        assert len(a) > 0 and a.tobytes() == b.tobytes() == c.tobytes()


def test_batch_instances_defer_too_and_ext_calls_end_a_run(vk, monkeypatch):
    w, h = 320, 240
    imgs = _images(vk, 16, w, h, 8100)
    with _instance(vk, monkeypatch, False, w * h, 32, batch_capacity=16) as inst:
        inst.detectFeaturesBatch(imgs, 0)
        inst.detectFeaturesBatch(imgs[::-1], 16)
        ref = [inst.downloadFeatures(i) for i in range(16)]
        inst.matchFeaturesBatch(list(range(16)), list(range(16, 32)))
        mref = [inst.downloadMatchesBatch(k) for k in range(16)]
    with _instance(vk, monkeypatch, True, w * h, 32, batch_capacity=16) as inst:
        for i, img in enumerate(imgs):
            inst.detectFeatures(img, 16 + i)
        inst.detectFeaturesBatch(imgs, 0)  # launches what is staged first: both land in stream order
        for i in range(16):
            assert inst.downloadFeatures(i).tobytes() == ref[i].tobytes()
            assert inst.downloadFeatures(16 + i).tobytes() == ref[i].tobytes()
        assert inst.getDeferredStats() == (1, 15)
        # the batched matching of deferred detections
        for i, img in enumerate(imgs):
            inst.detectFeatures(img, i)
        for i, img in enumerate(imgs[::-1]):
            inst.detectFeatures(img, 16 + i)
        inst.matchFeaturesBatch(list(range(16)), list(range(16, 32)))
        for k in range(16):
            assert inst.downloadMatchesBatch(k).tobytes() == mref[k].tobytes()


def test_small_defer_max_and_profiling(vk, monkeypatch):
    w, h = 200, 152
    imgs = _images(vk, 10, w, h, 42)
    with _instance(vk, monkeypatch, False, w * h, 10) as inst:
        ref = []
        for i, img in enumerate(imgs):
            inst.detectFeatures(img, i)
            ref.append(inst.downloadFeatures(i))
    with _instance(vk, monkeypatch, True, w * h, 10, defer_max=3) as inst:
        for rep in range(3):
            for i, img in enumerate(imgs):
                inst.detectFeatures(img, i)
            for i in range(10):
                assert inst.downloadFeatures(i).tobytes() == ref[i].tobytes()
        # a profiled instance times every call on its own: nothing is staged
        before = inst.getDeferredStats()
        inst.setProfiling(True)
        for i, img in enumerate(imgs):
            inst.detectFeatures(img, i)
        assert inst.getAccumulatedDetectTimings()["nb_calls"] == 10
        assert inst.getDeferredStats() == before
        for i in range(10):
            assert inst.downloadFeatures(i).tobytes() == ref[i].tobytes()
