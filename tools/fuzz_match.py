#!/usr/bin/env python3
"""Extra randomised 2-NN / filtered-match parity against the oracle with fresh seeds: random sizes over all kernel regimes
(one-launch small kernel, stream-decomposed single pair, batched 16 / 32 rows per wave), low-entropy descriptors (ties), duplicates, batched pairs.
usage (on the GPU box): python tools/fuzz_match.py <seed> <cases> [max_n]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vulkansift_amd import api as vk
from oracle import oracle
vk.lib().vksift_setLogLevel(vk.VKSIFT_LOG_ERROR)
seed, cases = int(sys.argv[1]), int(sys.argv[2])
max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 9000
rng = np.random.default_rng(seed)


def descs(n, mode):
    if mode == 0:
        return vk.gen_synthetic_descriptors(int(rng.integers(1, 1 << 30)), n)
    if mode == 1:                                   # few distinct values: many exact ties
        return rng.integers(0, 3, (n, 128)).astype(np.uint8) * 40
    if mode == 2:                                   # extremes: largest distances, the float-sqrt collision range
        return (rng.integers(0, 2, (n, 128)) * 255).astype(np.uint8)
    base = vk.gen_synthetic_descriptors(int(rng.integers(1, 1 << 30)), max(1, n // 7))
    return np.clip(base[rng.integers(0, len(base), n)].astype(np.int32) + rng.integers(-2, 3, (n, 128)), 0, 255).astype(np.uint8)


def feats(d):
    f = np.zeros(len(d), vk.FEATURE_DTYPE)
    f["descriptor"] = d
    return f


bad = 0
for case in range(cases):
    npairs = int(rng.choice([1, 1, 2, 5, 9]))
    big = rng.random() < 0.25
    sizes = [(int(rng.integers(2, max_n if big else 2600)), int(rng.integers(2, max_n if big else 2600))) for _ in range(npairs)]
    sets = [(descs(na, int(rng.integers(0, 4))), descs(nb, int(rng.integers(0, 4)))) for na, nb in sizes]
    cap = max(max(len(a), len(b)) for a, b in sets)
    cfg = vk.default_config(max_nb_sift_per_buffer=max(cap, 1000), sift_buffer_count=2 * npairs)
    with vk.Instance(cfg, batch_capacity=max(1, npairs)) as inst:
        for k, (a, b) in enumerate(sets):
            inst.uploadFeatures(feats(a), 2 * k)
            inst.uploadFeatures(feats(b), 2 * k + 1)
        if npairs == 1:
            inst.matchFeatures(0, 1)
            got = [inst.downloadMatches()]
        else:
            inst.matchFeaturesBatch([2 * k for k in range(npairs)], [2 * k + 1 for k in range(npairs)])
            got = [inst.downloadMatchesBatch(k) for k in range(npairs)]
        inst.matchFeaturesFiltered([2 * k for k in range(npairs)], [2 * k + 1 for k in range(npairs)], 0.8, True)
        filt = [inst.downloadFilteredMatches(k) for k in range(npairs)]
    for k, (a, b) in enumerate(sets):
        ref = oracle.match_2nn(a, b)
        ok = len(got[k]) == len(ref) and got[k].tobytes() == ref.tobytes()
        ra, rbm = oracle.filter_matches(ref, oracle.match_2nn(b, a), 0.8, True)
        ok = ok and np.array_equal(filt[k]["idx_a"], ra) and np.array_equal(filt[k]["idx_b"], rbm)
        if not ok:
            bad += 1
            print("MISMATCH case", case, "pair", k, sizes[k])
print("cases", cases, "bad", bad)
sys.exit(1 if bad else 0)
