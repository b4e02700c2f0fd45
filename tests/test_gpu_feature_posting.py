"""Feature posting (vksift_internal.h: h_post): a single-image detection leaves its dense records in pinned memory and
vksift_downloadFeatures copies them out. The records must be byte-identical to the ones the copy paths deliver
(VKSIFT_POST_FEATURES=0), whatever the order of detections, buffers and downloads."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _instance(vk, monkeypatch, post, w, h, nbuf=4, **kw):
    monkeypatch.setenv("VKSIFT_POST_FEATURES", "1" if post else "0")
    return vk.Instance(vk.default_config(input_image_max_size=w * h, sift_buffer_count=nbuf, **kw))


@pytest.mark.parametrize("w,h", [(640, 480), (1536, 1024), (97, 61)])
def test_posted_records_equal_copied_records(vk, monkeypatch, w, h):
    imgs = [vk.gen_synthetic_image_family(700 + i + w, w, h, i % 3) for i in range(4)]
    with _instance(vk, monkeypatch, False, w, h) as inst:
        ref = []
        for i, img in enumerate(imgs):
            inst.detectFeatures(img, i)
            ref.append(inst.downloadFeatures(i))
    assert sum(len(r) for r in ref) > 0
    with _instance(vk, monkeypatch, True, w, h) as inst:
        # plain protocol, twice per buffer (the second replay of a captured sequence included)
        for rep in range(3):
            for i, img in enumerate(imgs):
                inst.detectFeatures(img, i)
                assert inst.downloadFeatures(i).tobytes() == ref[i].tobytes()
                assert inst.downloadFeatures(i).tobytes() == ref[i].tobytes()  # a second download of the same buffer
        # two detections in flight on buffers of different slots, fetched afterwards in both orders
        inst.detectFeatures(imgs[0], 0)
        inst.detectFeatures(imgs[1], 1)
        assert inst.downloadFeatures(1).tobytes() == ref[1].tobytes()
        assert inst.downloadFeatures(0).tobytes() == ref[0].tobytes()
        # ... and on buffers that share a slot: the older one comes through the copy path
        inst.detectFeatures(imgs[0], 0)
        inst.detectFeatures(imgs[2], 2)
        assert inst.downloadFeatures(0).tobytes() == ref[0].tobytes()
        assert inst.downloadFeatures(2).tobytes() == ref[2].tobytes()
        # a buffer refilled from the host is no longer the posted one
        inst.detectFeatures(imgs[3], 3)
        inst.uploadFeatures(ref[1], 3)
        assert inst.downloadFeatures(3).tobytes() == ref[1].tobytes()
        # the same buffer detected with another image
        inst.detectFeatures(imgs[2], 1)
        assert inst.downloadFeatures(1).tobytes() == ref[2].tobytes()


def test_posting_switches_off_when_nobody_fetches_and_on_again(vk, monkeypatch):
    w, h = 320, 240
    img = vk.gen_synthetic_image(4242, w, h)
    with _instance(vk, monkeypatch, True, w, h) as inst:
        inst.detectFeatures(img, 0)
        ref = inst.downloadFeatures(0)
        for _ in range(40):  # detections whose features stay on the device (matched only, say)
            inst.detectFeatures(img, 0)
            assert inst.getFeaturesNumber(0) == len(ref)
        assert inst.downloadFeatures(0).tobytes() == ref.tobytes()  # copy path; re-arms the posting
        for _ in range(3):
            inst.detectFeatures(img, 0)
            assert inst.downloadFeatures(0).tobytes() == ref.tobytes()


def test_posting_with_more_features_than_the_buffer_holds(vk, monkeypatch):
    """sections capped by max_nb_sift_per_buffer: the posted records are the stored ones"""
    w, h = 640, 480
    img = vk.gen_synthetic_image(99, w, h)
    out = []
    for post in (False, True):
        with _instance(vk, monkeypatch, post, w, h, max_nb_sift_per_buffer=300) as inst:
            inst.detectFeatures(img, 0)
            out.append(inst.downloadFeatures(0))
    assert 0 < len(out[0]) <= 300 and out[0].tobytes() == out[1].tobytes()
