"""The matcher for reference sets beyond 32768 rows (k_match_scan32 -> k_match_fix -> k_match_redo_rows, match.hip): a branch-free
scan keeps three cells (32-column sub-blocks of a lane) per lane, the exact finish recomputes the distances of the columns in the
cells that reach the row's threshold, rows that could have lost a tied cell or whose distances leave the float-exact range are
replayed. Each case against the oracle, bit for bit, through the device-pointer entry:
  * SIFT-like rows with exact hits in the first and the last column, Q7 (b0 == b1)
  * duplicated reference rows everywhere: equal cell maxima in every lane -> two-way ties are evaluated, deeper ties replayed
  * one reference row repeated 5000 times: every lane's four best cells tie -> every row takes the replay kernel
  * full-range random bytes (d2 up to 2^23: quirk Q8, the float-sqrt order) -> every row takes the replay kernel
  * a query shard with a_index_base != 0 (what the sharded matcher passes)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _match(vk, a, b, base=0):
    import torch
    from vulkansift_amd import multigpu

    rec = multigpu.hip_match_fn(torch.from_numpy(a).cuda(), base, torch.from_numpy(b).cuda())
    torch.cuda.synchronize()
    return multigpu.records_to_struct(rec.cpu().numpy())


def _check(got, ref, base=0):
    assert np.array_equal(got["idx_a"], ref["idx_a"] + base)
    for name in ("idx_b1", "idx_b2"):
        assert np.array_equal(got[name], ref[name]), (name, np.flatnonzero(got[name] != ref[name])[:10])
    for name in ("dist_a_b1", "dist_a_b2"):
        assert np.array_equal(got[name].view(np.uint32), ref[name].view(np.uint32)), name


def test_sift_like_rows_hits_at_both_ends_and_q7(vk, oracle):
    nb = 40003
    a = vk.gen_synthetic_descriptors(601, 700)
    b = vk.gen_synthetic_descriptors(602, nb)
    b[1] = b[0]
    a[0] = b[0]
    a[1] = b[nb - 1]
    a[2] = b[nb - 2]
    a[3] = b[32767]
    a[4] = b[32768]
    got = _match(vk, a, b, base=1234)
    ref = oracle.match_2nn(a, b)
    _check(got, ref, base=1234)
    assert got["idx_b1"][0] == 1 and got["idx_b2"][0] == 0 and got["idx_b1"][1] == nb - 1


def test_duplicated_reference_rows(vk, oracle):
    rng = np.random.default_rng(5)
    nb = 36000
    a = vk.gen_synthetic_descriptors(603, 500)
    b = vk.gen_synthetic_descriptors(604, nb)
    dup = rng.permutation(np.arange(2, nb))[: nb // 3]
    b[dup] = b[rng.integers(2, nb, len(dup))]
    a[::7] = b[rng.integers(0, nb, len(a[::7]))]
    _check(_match(vk, a, b), oracle.match_2nn(a, b))


def test_one_row_repeated_everywhere_takes_the_replay(vk, oracle):
    nb = 34000
    a = vk.gen_synthetic_descriptors(605, 300)
    b = vk.gen_synthetic_descriptors(606, nb)
    b[::7] = b[3]                      # ~4900 identical rows: the best cells of every lane tie
    a[:50] = b[3]
    _check(_match(vk, a, b), oracle.match_2nn(a, b))


def test_full_range_bytes_float_collisions(vk, oracle):
    rng = np.random.default_rng(6)
    a = rng.integers(0, 256, (260, 128), dtype=np.uint8)
    b = rng.integers(0, 256, (33000, 128), dtype=np.uint8)
    a[:32] = np.where(rng.random((32, 128)) < 0.5, 0, 255).astype(np.uint8)
    b[:2000] = np.where(rng.random((2000, 128)) < 0.5, 0, 255).astype(np.uint8)
    _check(_match(vk, a, b), oracle.match_2nn(a, b))


def test_scan_equals_the_pruning_kernel(vk):
    """VKSIFT_MATCH_SCAN=0 (the stream-decomposed pruning kernel of rounds 2-3) on the same inputs: identical records"""
    import os
    import subprocess
    import sys

    code = ("import sys, zlib, torch; sys.path.insert(0, %r); from vulkansift_amd import api, multigpu\n"
            "a = torch.from_numpy(api.gen_synthetic_descriptors(11, 3000)).cuda(); b = torch.from_numpy(api.gen_synthetic_descriptors(12, 45000)).cuda()\n"
            "r = multigpu.hip_match_fn(a, 7, b); torch.cuda.synchronize(); print('CRC', zlib.crc32(r.cpu().numpy().tobytes()))\n") % os.path.dirname(
        os.path.dirname(os.path.abspath(__file__)))
    crcs = []
    for scan in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, VKSIFT_MATCH_SCAN=scan))
        assert r.returncode == 0, r.stderr[-1500:]
        crcs.append([ln for ln in r.stdout.splitlines() if ln.startswith("CRC")][-1])
    assert crcs[0] == crcs[1]
