"""SURVEY.md 8(b)'s in-process form of the drop-in: several vksift instances in ONE process, each driven by its own host thread
(the reference: one GPU per instance, include/vulkansift/vulkansift.h:32-34 of the reference — an application with 8 GPUs creates 8
instances). One GPU is all a test box has, so both instances sit on device 0; what is under test is the host side — per-instance
streams, events, staging, the shared staging-thread pool, the lazily initialised process-wide state of the launch shims — under real
concurrency (ctypes releases the GIL for the duration of every call). Every result must equal the serial run byte for byte."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _batch_worker(vk, w, h, sets, out, idx, barrier, err):
    try:
        with vk.Instance(vk.default_config(input_image_max_size=w * h, sift_buffer_count=8), batch_capacity=8) as inst:
            barrier.wait()
            res = []
            for imgs in sets:
                inst.detectFeaturesBatch(imgs, 0)
                inst.matchFeaturesBatch(list(range(8)), [(i + 1) % 8 for i in range(8)])
                feats = [inst.downloadFeatures(i).tobytes() for i in range(8)]
                recs = [inst.downloadMatchesBatch(i).tobytes() for i in range(8)]
                res.append((feats, recs))
            out[idx] = res
    except Exception as e:  # noqa: BLE001
        err.append((idx, repr(e)))
        try:
            barrier.abort()
        except Exception:  # noqa: BLE001
            pass


def _plain_worker(vk, w, h, imgs, out, idx, barrier, err):
    """the reference's own calling pattern (one image per call, two buffers), 12 rounds"""
    try:
        with vk.Instance(vk.default_config(input_image_max_size=w * h, sift_buffer_count=2)) as inst:
            barrier.wait()
            res = []
            for r in range(12):
                a, b = imgs[r % len(imgs)], imgs[(r + 1) % len(imgs)]
                inst.detectFeatures(a, 0)
                inst.detectFeatures(b, 1)
                fa = inst.downloadFeatures(0).tobytes()
                inst.matchFeatures(0, 1)
                res.append((fa, inst.downloadFeatures(1).tobytes(), inst.downloadMatches().tobytes()))
            out[idx] = res
    except Exception as e:  # noqa: BLE001
        err.append((idx, repr(e)))
        try:
            barrier.abort()
        except Exception:  # noqa: BLE001
            pass


def _run(target, argsets, concurrent):
    n = len(argsets)
    out, err = [None] * n, []
    barrier = threading.Barrier(n if concurrent else 1)
    if concurrent:
        th = [threading.Thread(target=target, args=a + (out, i, barrier, err)) for i, a in enumerate(argsets)]
        for t in th:
            t.start()
        for t in th:
            t.join(600)
            assert not t.is_alive()
    else:
        for i, a in enumerate(argsets):
            target(*(a + (out, i, barrier, err)))
    assert not err, err
    return out


def test_two_batch_instances_on_two_threads_equal_the_serial_run(vk):
    w, h = 320, 240
    sets = [[[vk.gen_synthetic_image_family(5000 + 100 * t + 10 * s + i, w, h, (i + s + t) % 3) for i in range(8)] for s in range(5)] for t in range(2)]
    args = [(vk, w, h, sets[t]) for t in range(2)]
    serial = _run(_batch_worker, args, False)
    both = _run(_batch_worker, args, True)
    assert serial == both
    assert sum(len(f) for f in serial[0][0][0]) > 0 and serial[0] != serial[1]


def test_plain_and_batch_instances_side_by_side(vk):
    """three threads: two plain single-image instances (hipGraph replay, feature posting, deferred submission) and a batch instance"""
    w, h = 640, 480
    imgs = [[vk.gen_synthetic_image(6000 + 10 * t + i, w, h) for i in range(3)] for t in range(2)]
    bsets = [[vk.gen_synthetic_image(6100 + 10 * s + i, 320, 240) for i in range(8)] for s in range(6)]
    serial = [_run(_plain_worker, [(vk, w, h, imgs[t])], False)[0] for t in range(2)] + [_run(_batch_worker, [(vk, 320, 240, bsets)], False)[0]]
    out, err = [None] * 3, []
    barrier = threading.Barrier(3)
    th = [threading.Thread(target=_plain_worker, args=(vk, w, h, imgs[t], out, t, barrier, err)) for t in range(2)]
    th.append(threading.Thread(target=_batch_worker, args=(vk, 320, 240, bsets, out, 2, barrier, err)))
    for t in th:
        t.start()
    for t in th:
        t.join(600)
        assert not t.is_alive()
    assert not err, err
    assert out == serial
