"""k_match_pk — the branch-free packed-key matcher that takes the pairs of a batched matching whose reference set has at most 4096
rows (match.hip). One vksift_ext_matchFeaturesBatch call mixes every case its exactness argument has to survive, each pair compared
with the oracle bit for bit:
  * SIFT-like rows incl. a self-match (d2 = 0: the largest key field) and duplicated rows (equal d2: the earlier index wins)
  * quirk Q7 (d(b0) == d(b1)) alone, with a third equal column, and with a closer column elsewhere
  * reference sets of exactly 4096 rows with the best match in the LAST row (index field 0), of 4097 rows (not this kernel's: the
    pruning kernels of the same launch sequence must take the slot) and of 2 and 3 rows (one partial tile)
  * full-range random bytes: d2 up to 2^23 does not fit the 20-bit key field, the keys wrap and may look closer than they are —
    the verification step has to send exactly those rows to the scalar replay (incl. rows whose float sqrt collides, quirk Q8)
  * all-255 against all-0 rows: every d2 = 128 * 255^2 (wrapped AND tied everywhere)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _feats(vk, desc):
    f = np.zeros(len(desc), vk.FEATURE_DTYPE)
    f["descriptor"] = desc
    return f


def _cases(vk):
    rng = np.random.default_rng(77)
    out = []
    a = vk.gen_synthetic_descriptors(501, 1900)
    out.append(("self-match", a, a.copy()))
    b = vk.gen_synthetic_descriptors(502, 2000)
    b[1] = b[0]
    b[rng.permutation(np.arange(2, 2000))[:500]] = b[rng.integers(2, 2000, 500)]
    a2 = vk.gen_synthetic_descriptors(503, 1500)
    a2[::9] = b[rng.integers(0, 2000, len(a2[::9]))]
    a2[::97] = b[0]
    out.append(("duplicates + Q7", a2, b))
    b3 = vk.gen_synthetic_descriptors(504, 300)
    b3[1] = b3[0]
    b3[7] = b3[0]                                   # a third column equal to b0 == b1
    a3 = vk.gen_synthetic_descriptors(505, 200)
    a3[5] = b3[0]
    a3[6] = b3[100]                                 # closer column elsewhere while b0 == b1
    out.append(("Q7 with a third equal column", a3, b3))
    b4 = vk.gen_synthetic_descriptors(506, 4096)
    a4 = vk.gen_synthetic_descriptors(507, 700)
    a4[3] = b4[4095]
    a4[4] = b4[4094]
    out.append(("4096 rows, best match in the last row", a4, b4))
    b5 = vk.gen_synthetic_descriptors(508, 4097)
    a5 = vk.gen_synthetic_descriptors(509, 600)
    a5[3] = b5[4096]
    out.append(("4097 rows: the pruning kernels' slot", a5, b5))
    out.append(("2 rows", vk.gen_synthetic_descriptors(510, 70), vk.gen_synthetic_descriptors(511, 2)))
    out.append(("3 rows, 1 query", vk.gen_synthetic_descriptors(512, 1), vk.gen_synthetic_descriptors(513, 3)))
    a6 = rng.integers(0, 256, (600, 128), dtype=np.uint8)
    b6 = rng.integers(0, 256, (3000, 128), dtype=np.uint8)
    a6[:64] = np.where(rng.random((64, 128)) < 0.5, 0, 255).astype(np.uint8)
    b6[:512] = np.where(rng.random((512, 128)) < 0.5, 0, 255).astype(np.uint8)
    a6[100:140] = vk.gen_synthetic_descriptors(514, 40)          # a few rows with small norms among the large ones
    out.append(("full-range bytes (keys wrap, verification + replay)", a6, b6))
    a7 = np.full((40, 128), 255, np.uint8)
    b7 = np.zeros((130, 128), np.uint8)
    out.append(("all d2 = 128 * 255^2", a7, b7))
    return out


def test_batched_pairs_of_every_kind_equal_the_oracle(vk, oracle):
    cases = _cases(vk)
    n = len(cases)
    cfg = vk.default_config(sift_buffer_count=2 * n, max_nb_sift_per_buffer=4200)
    with vk.Instance(cfg, batch_capacity=n) as inst:
        for i, (_, a, b) in enumerate(cases):
            inst.uploadFeatures(_feats(vk, a), 2 * i)
            inst.uploadFeatures(_feats(vk, b), 2 * i + 1)
        inst.matchFeaturesBatch([2 * i for i in range(n)], [2 * i + 1 for i in range(n)])
        got = [inst.downloadMatchesBatch(i) for i in range(n)]
        # both directions of every pair in a second call (the cache entries are reused, N_A and N_B swap roles)
        inst.matchFeaturesBatch([2 * i + 1 for i in range(n)], [2 * i for i in range(n)])
        rev = [inst.downloadMatchesBatch(i) for i in range(n)]
    for (name, a, b), m, r in zip(cases, got, rev):
        ref = oracle.match_2nn(a, b)
        assert len(m) == len(a), name
        assert m.tobytes() == ref.tobytes(), (name, np.flatnonzero(m["idx_b1"] != ref["idx_b1"])[:8], np.flatnonzero(m["idx_b2"] != ref["idx_b2"])[:8])
        if len(a) >= 2:
            assert r.tobytes() == oracle.match_2nn(b, a).tobytes(), name + " (reverse)"


def test_packed_kernel_equals_the_pruning_kernels(vk, monkeypatch):
    """the same batched self-matching with VKSIFT_MATCH_PK=0 (pruning kernels only): identical records"""
    import os
    import subprocess
    import sys

    code = ("import sys, zlib, numpy as np; sys.path.insert(0, %r); from vulkansift_amd import api as vk\n"
            "imgs = [vk.gen_synthetic_image(900 + i, 480, 360) for i in range(8)]\n"
            "cfg = vk.default_config(sift_buffer_count=8)\n"
            "inst = vk.Instance(cfg, batch_capacity=8)\n"
            "inst.detectFeaturesBatch(imgs, 0)\n"
            "inst.matchFeaturesBatch(list(range(8)), [(i + 1) %% 8 for i in range(8)])\n"
            "crc = 0\n"
            "for i in range(8): crc = zlib.crc32(inst.downloadMatchesBatch(i).tobytes(), crc)\n"
            "print('CRC', crc)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    crcs = []
    for pk in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, VKSIFT_MATCH_PK=pk))
        assert r.returncode == 0, r.stderr[-1500:]
        crcs.append([ln for ln in r.stdout.splitlines() if ln.startswith("CRC")][-1])
    assert crcs[0] == crcs[1]


def test_batch_with_reference_sets_beyond_32k_rows_and_counts_still_on_the_device(vk):
    """A batched matching queued right behind its detection: the counts are on the device, so the host cannot know whether a slot lies beyond
    the packed-key kernel's usual range (32 768 reference rows) and queues no pruning-kernel grid for it — the packed-key kernel serves
    every slot, walking B in super-chunks of 4096 columns. The same matching again, once the counts have reached the host (then the pruning
    kernels take the slots beyond the range), and the single-pair path must give the same records."""
    w, h = 3456, 2304
    imgs = [vk.gen_synthetic_image(31 + i, w, h) for i in range(2)]
    with vk.Instance(vk.default_config(input_image_max_size=w * h, sift_buffer_count=2, max_nb_sift_per_buffer=100000), batch_capacity=2) as inst:
        inst.detectFeaturesBatch(imgs, 0)
        inst.matchFeaturesBatch([0, 1], [1, 0])                   # counts unknown: packed-key kernel for both slots
        first = [inst.downloadMatchesBatch(k) for k in range(2)]
        n = [inst.getFeaturesNumber(i) for i in range(2)]
        assert min(n) > 32768, n                                  # the case under test
        inst.matchFeaturesBatch([0, 1], [1, 0])                   # counts known and beyond the range: pruning kernels
        second = [inst.downloadMatchesBatch(k) for k in range(2)]
        inst.matchFeatures(0, 1)                                  # single pair: stream decomposition
        single = inst.downloadMatches()
    assert [len(m) for m in first] == n
    for a, b in zip(first, second):
        assert a.tobytes() == b.tobytes()
    assert first[0].tobytes() == single.tobytes()
