"""The packed-key matcher's top-2 fold (vulkansift_amd/csrc/hip/match.hip: k_match_pk) restated in Python: four candidates per chain
are folded with five operations — t_a = med3(K1, a, b), K1' = max3(K1, a, b), t_b = med3(K1', c, d), K1'' = max3(K1', c, d),
K2' = max3(K2, t_a, t_b) — and the four chains of a lane are merged after the scan. The result must be the two largest keys of the
multiset whatever the arrival order and whatever duplicates it holds (CPU test of the identity the kernel relies on; the kernel
itself is checked bit for bit by tests/test_gpu_match_packed.py)."""
import itertools
import random


def med3(a, b, c):
    return max(min(a, b), min(max(a, b), c))


def fold4(k1, k2, a, b, c, d):
    ta = med3(k1, a, b)
    ka = max(k1, a, b)
    tb = med3(ka, c, d)
    return max(ka, c, d), max(k2, ta, tb)


def scan(keys):
    """keys: a multiple of 16 per sub-block, dealt to four chains as the kernel does (key i, i + 4, i + 8, i + 12 -> chain i)"""
    k1, k2 = [0] * 4, [0] * 4
    for s in range(0, len(keys), 16):
        sub = keys[s:s + 16]
        for c in range(4):
            k1[c], k2[c] = fold4(k1[c], k2[c], sub[c], sub[c + 4], sub[c + 8], sub[c + 12])
    kb, ks = k1[0], k2[0]
    for c in range(1, 4):
        ks = max(min(kb, k1[c]), max(ks, k2[c]))
        kb = max(kb, k1[c])
    return kb, ks


def test_fold4_is_the_top2_of_the_multiset_exhaustive_small():
    # every multiset of (K1 >= K2) + four keys over a small alphabet, duplicates included
    vals = range(0, 5)
    for k1, k2 in ((x, y) for x in vals for y in vals if y <= x):
        for quad in itertools.product(vals, repeat=4):
            want = sorted((k1, k2) + quad, reverse=True)[:2]
            assert list(fold4(k1, k2, *quad)) == want, (k1, k2, quad)


def test_scan_matches_sorted_top2_random_and_adversarial():
    rng = random.Random(20260929)
    for trial in range(300):
        n = 16 * rng.randint(1, 12)
        mode = trial % 4
        if mode == 0:
            keys = [rng.getrandbits(32) for _ in range(n)]
        elif mode == 1:  # many equal keys (forced zeros of columns beyond B, equal distances with different index fields)
            keys = [rng.choice((0, 0, 4097, 1 << 31, (1 << 31) + 1)) for _ in range(n)]
        elif mode == 2:  # ascending / descending runs
            keys = sorted(rng.getrandbits(32) for _ in range(n))
            if trial & 4:
                keys.reverse()
        else:            # the two best in the same chain, in the same group of four
            keys = [rng.getrandbits(20) for _ in range(n)]
            s = 16 * rng.randrange(n // 16)
            c = rng.randrange(4)
            keys[s + c], keys[s + c + 8] = (1 << 32) - 1, (1 << 32) - 2
        want = sorted(keys + [0, 0], reverse=True)[:2]
        assert list(scan(keys)) == want
