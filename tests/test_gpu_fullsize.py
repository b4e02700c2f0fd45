"""BASELINE config 4 at its full size (50 000 x 50 000 x 128-D) against an implementation that shares nothing with the HIP matcher or
the oracle: exact squared distances from a torch fp32 GEMM (all operands are integers below 2^24, so every partial sum is exact),
the eight smallest (d2, index) candidates per row by an integer key, and the reference's scan semantics
(Get2NearestNeighbors.comp:43-96: float-sqrt distances, strict '<' in index order, the b0/b1 initialisation of quirk Q7) applied to
those candidates in numpy. The oracle needs minutes for this size; this check needs seconds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _reference_top2(a, b, chunk=2000, k=8):
    import torch
    dev = torch.device("cuda", 0)
    ta = torch.from_numpy(a.astype(np.float32)).to(dev)
    tb = torch.from_numpy(b.astype(np.float32)).to(dev)
    na2 = (ta * ta).sum(1).to(torch.int64)
    nb2 = (tb * tb).sum(1).to(torch.int64)
    n, m = len(a), len(b)
    cand_d2 = np.empty((n, k), np.int64)
    cand_ix = np.empty((n, k), np.int64)
    d01 = np.empty((n, 2), np.int64)
    for i0 in range(0, n, chunk):
        g = ta[i0:i0 + chunk] @ tb.T                                     # exact: integers, |sum| < 2^24
        d2 = na2[i0:i0 + chunk, None] + nb2[None, :] - 2 * g.to(torch.int64)
        d01[i0:i0 + chunk] = d2[:, :2].cpu().numpy()
        key = d2 * (1 << 17) + torch.arange(m, device=dev, dtype=torch.int64)[None, :]
        top = torch.topk(key, k, dim=1, largest=False).values           # unique keys: no tie ambiguity
        cand_d2[i0:i0 + chunk] = (top >> 17).cpu().numpy()
        cand_ix[i0:i0 + chunk] = (top & ((1 << 17) - 1)).cpu().numpy()
    dist = np.sqrt(cand_d2.astype(np.float32)).astype(np.float32)        # numpy: correctly rounded float sqrt
    assert (dist[:, k - 1] > dist[:, 1]).all(), "candidate list too short for this data"
    tie01 = np.sqrt(d01[:, 0].astype(np.float32)) == np.sqrt(d01[:, 1].astype(np.float32))
    # order of the scan: (float distance, index), with indices 0 and 1 swapped when d(b0) == d(b1) (quirk Q7)
    ixk = np.where((cand_ix < 2) & tie01[:, None], cand_ix ^ 1, cand_ix)
    order = np.lexsort((ixk, dist), axis=1)[:, :2]
    rows = np.arange(n)
    return cand_ix[rows, order[:, 0]], cand_ix[rows, order[:, 1]], dist[rows, order[:, 0]], dist[rows, order[:, 1]]


@pytest.mark.parametrize("via_instance", [False, True])
def test_c4_50k_by_50k_equals_gemm_reference(vk, via_instance):
    import torch
    from vulkansift_amd import multigpu
    n = 50000
    a = vk.gen_synthetic_descriptors(1, n)                               # bench.py's config-4 inputs
    b = vk.gen_synthetic_descriptors(2, n)
    b[1] = b[0]                                                          # quirk Q7 on every row
    b[40000] = b[77]                                                     # a duplicate far away: the earlier index wins
    a[123] = b[77]
    i1, i2, d1, d2 = _reference_top2(a, b)
    if via_instance:
        fa = np.zeros(n, vk.FEATURE_DTYPE)
        fb = np.zeros(n, vk.FEATURE_DTYPE)
        fa["descriptor"], fb["descriptor"] = a, b
        with vk.Instance(vk.default_config(max_nb_sift_per_buffer=n)) as inst:
            inst.uploadFeatures(fa, 0)
            inst.uploadFeatures(fb, 1)
            inst.matchFeatures(0, 1)
            got = inst.downloadMatches()
    else:
        rec = multigpu.hip_match_fn(torch.from_numpy(a).cuda(), 0, torch.from_numpy(b).cuda())
        torch.cuda.synchronize()
        got = multigpu.records_to_struct(rec.cpu().numpy())
    assert np.array_equal(got["idx_a"], np.arange(n))
    assert np.array_equal(got["idx_b1"], i1), np.flatnonzero(got["idx_b1"] != i1)[:10]
    assert np.array_equal(got["idx_b2"], i2), np.flatnonzero(got["idx_b2"] != i2)[:10]
    assert np.array_equal(got["dist_a_b1"].view(np.uint32), d1.view(np.uint32))
    assert np.array_equal(got["dist_a_b2"].view(np.uint32), d2.view(np.uint32))
    assert got["idx_b1"][123] == 77 and got["idx_b2"][123] == 40000 and got["dist_a_b1"][123] == 0


@pytest.mark.parametrize("ups", [True, False])
def test_c3_1080p_scale_space_equals_numpy_restatement(vk, ups):
    """BASELINE config 3's image size: every Gaussian and DoG plane of every octave of a 1920x1080 frame against the numpy
    restatement of the dense stages (tests/np_restatement.py: sampler-fetch formulation, closed-form up-sampling, odd-texel
    down-sampling — written independently of the oracle and of the kernels), tolerance 3e-6 on texels in [0, 1]"""
    import np_restatement as R
    img = vk.gen_synthetic_image(4242, 1920, 1080)
    ref = R.build_pyramid(img, ups=ups)
    cfg = vk.default_config(use_input_upsampling=ups, input_image_max_size=1920 * 1080)
    with vk.Instance(cfg) as inst:
        inst.detectFeatures(img, 0)
        assert inst.getScaleSpaceNbOctaves() == len(ref) == (7 if ups else 6)
        worst_g = worst_d = 0.0
        for o, (g, d) in enumerate(ref):
            assert inst.getScaleSpaceOctaveResolution(o) == (g.shape[2], g.shape[1])
            for s in range(g.shape[0]):
                worst_g = max(worst_g, float(np.abs(inst.downloadScaleSpaceImage(o, s) - g[s]).max()))
            for s in range(d.shape[0]):
                worst_d = max(worst_d, float(np.abs(inst.downloadDoGImage(o, s) - d[s]).max()))
    assert worst_g < 3e-6 and worst_d < 3e-6, (worst_g, worst_d)
