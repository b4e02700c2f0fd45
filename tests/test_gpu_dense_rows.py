"""The matcher's view of a SIFT buffer written by the descriptor launch itself (vksift_hip_DenseRows, include/vksift_hip.h): once an
instance has matched, a detection leaves the dense descriptor rows, their shifted norms and the row count of every buffer behind
itself and the gather pass (pack_BufferMemory's counterpart, sift_memory.c:957-1047) is not queued for them. Everything the matcher and
the descriptor export deliver must be byte-identical to the gather path (VKSIFT_TUNE_DENSE_ROWS = 1), and the exported rows must be the
descriptor bytes of the downloaded records, in download order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TUNE_DENSE_ROWS = 7


def _export(inst, buf, n):
    """vksift_ext_exportDescriptorsDevice into device memory of the library's own allocator (no torch: the file also runs under the
    sanitizer build, where torch does not load)"""
    import ctypes as C
    from vulkansift_amd import api

    L = api.lib()
    L.vksift_hip_malloc.restype = C.c_void_p
    L.vksift_hip_malloc.argtypes = [C.c_size_t]
    L.vksift_hip_free.argtypes = [C.c_void_p]
    L.vksift_hip_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.vksift_hip_stream_sync.argtypes = [C.c_void_p]
    d = L.vksift_hip_malloc(max(n, 1) * 128)
    assert d
    try:
        assert inst.exportDescriptorsDevice(buf, d) == n
        out = np.zeros((max(n, 1), 128), np.uint8)
        assert L.vksift_hip_memcpy_d2h(out.ctypes.data, d, max(n, 1) * 128, None) == 0 and L.vksift_hip_stream_sync(None) == 0
    finally:
        L.vksift_hip_free(d)
    return out[:n]


def _blank(w, h):
    return np.zeros((h, w), np.uint8)


def _run_batch(vk, off, w, h, sets, pairs, batch, **cfg):
    """detect every image set into buffers 0.., match `pairs` after each; returns per set (features, exported rows, match records)"""
    vk.lib().vksift_hip_tune(TUNE_DENSE_ROWS, 1 if off else 0)
    out = []
    try:
        with vk.Instance(vk.default_config(input_image_max_size=w * h, sift_buffer_count=len(sets[0]), **cfg), batch_capacity=batch) as inst:
            for imgs in sets:
                if batch > 1:
                    inst.detectFeaturesBatch(imgs, 0)
                else:
                    for i, img in enumerate(imgs):
                        inst.detectFeatures(img, i)
                recs = []
                if batch > 1:
                    inst.matchFeaturesBatch([p[0] for p in pairs], [p[1] for p in pairs])
                    recs = [inst.downloadMatchesBatch(k).tobytes() for k in range(len(pairs))]
                else:
                    for a, b in pairs:
                        inst.matchFeatures(a, b)
                        recs.append(inst.downloadMatches().tobytes())
                feats = [inst.downloadFeatures(i) for i in range(len(imgs))]
                rows = [_export(inst, i, len(feats[i])) for i in range(len(imgs))]
                out.append((feats, rows, recs))
    finally:
        vk.lib().vksift_hip_tune(TUNE_DENSE_ROWS, 0)
    return out


def _check(vk, w, h, sets, pairs, batch, **cfg):
    ref = _run_batch(vk, True, w, h, sets, pairs, batch, **cfg)
    got = _run_batch(vk, False, w, h, sets, pairs, batch, **cfg)
    total = 0
    for (fr, rr, mr), (fg, rg, mg) in zip(ref, got):
        for i in range(len(fr)):
            assert fr[i].tobytes() == fg[i].tobytes()
            assert rr[i].tobytes() == rg[i].tobytes()
            # the rows ARE the descriptors of the downloaded records, in download order
            assert rg[i].tobytes() == np.ascontiguousarray(fg[i]["descriptor"]).tobytes()
            total += len(fg[i])
        assert mr == mg
    return total


@pytest.mark.parametrize("cfg", [{}, {"use_input_upsampling": False}, {"max_nb_sift_per_buffer": 300},
                                 {"pyramid_precision_mode": 1}, {"descriptor_format": 1, "max_nb_orientation_per_keypoint": 0}])
def test_batch_rows_equal_the_gather_pass(vk, cfg):
    w, h = 320, 240
    sets = [[vk.gen_synthetic_image_family(9100 + 10 * s + i, w, h, (i + s) % 3) for i in range(8)] for s in range(3)]
    sets[1][3] = _blank(w, h)                                # no feature at all: its two zero rows (quirk Q6) and a row count of 0
    sets[2][0] = vk.gen_synthetic_image(77, w, h, 1)         # a handful of features
    pairs = [(0, 0), (3, 3), (5, 5), (0, 3), (3, 0), (1, 2), (7, 3), (6, 6)]  # at most batch_capacity pairs per call
    assert _check(vk, w, h, sets, pairs, 8, **cfg) > 1000


def test_single_image_instance_and_graph_replay(vk):
    """the first detections run before the cache exists (and may be captured as a graph without the rows); every later one writes them"""
    w, h = 640, 480
    imgs = [vk.gen_synthetic_image(9300 + i, w, h) for i in range(2)]
    sets = [imgs, imgs[::-1], imgs, imgs, [_blank(w, h), imgs[0]], imgs]
    assert _check(vk, w, h, sets, [(0, 1), (1, 0), (0, 0)], 1) > 1000


def test_upload_and_resolution_change_fall_back_to_the_gather(vk):
    w, h = 320, 240
    a, b = vk.gen_synthetic_image(9401, w, h), vk.gen_synthetic_image(9402, w, h)
    small = vk.gen_synthetic_image(9403, 96, 64)
    res = []
    for off in (True, False):
        vk.lib().vksift_hip_tune(TUNE_DENSE_ROWS, 1 if off else 0)
        try:
            with vk.Instance(vk.default_config(input_image_max_size=w * h, sift_buffer_count=3)) as inst:
                r = []
                inst.detectFeatures(a, 0)
                inst.detectFeatures(b, 1)
                inst.matchFeatures(0, 1)
                r.append(inst.downloadMatches().tobytes())
                fa = inst.downloadFeatures(0)
                inst.detectFeatures(b, 0)          # dense rows of b in entry 0 ...
                inst.uploadFeatures(fa, 0)         # ... replaced by an upload: the entry is stale, the gather rebuilds it
                inst.matchFeatures(0, 1)
                r.append(inst.downloadMatches().tobytes())
                inst.detectFeatures(small, 2)      # fewer octaves, other section table
                inst.matchFeatures(2, 0)
                r.append(inst.downloadMatches().tobytes())
                inst.detectFeatures(a, 2)
                inst.matchFeatures(2, 0)
                r.append(inst.downloadMatches().tobytes())
                r.append(_export(inst, 2, inst.getFeaturesNumber(2)).tobytes())
                res.append(r)
        finally:
            vk.lib().vksift_hip_tune(TUNE_DENSE_ROWS, 0)
    assert res[0] == res[1]
    assert len(res[0][0]) > 0 and res[0][0] == res[0][1]  # the upload restored a's features: same records as the first matching
