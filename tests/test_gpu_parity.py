"""GPU parity: the HIP path (through the vksift_* C-ABI) against the CPU oracle.

The oracle runs in its "det" math mode (shared detmath.h) for bit-exact checks, and in libm mode for
the tolerance checks that bound how far any IEEE-ish exp/atan implementation may move the result.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfgs(vk, oracle, **kw):
    okw = {}
    vkw = {}
    for k, v in kw.items():
        if k in ("use_input_upsampling", "use_hardware_interpolated_blur"):
            okw[k] = int(v)
            vkw[k] = bool(v)
        elif k == "descriptor_format":
            okw["use_vlfeat_format"] = int(v)
            vkw[k] = int(v)
        else:
            okw[k] = v
            vkw[k] = v
    return vk.default_config(**vkw), oracle.default_config(math_mode=1, **okw)


@pytest.mark.parametrize("w,h,kw", [
    (320, 240, {}),
    (200, 150, {"use_input_upsampling": False}),
    (257, 131, {"use_hardware_interpolated_blur": False}),
    (368, 224, {"nb_scales_per_octave": 2}),          # largest kernels: 20 one-sided taps (radius 19 > two 8-row groups)
    (368, 224, {"nb_scales_per_octave": 1, "seed_scale_sigma": 2.4}),
])
def test_pyramid_bit_exact(vk, oracle, w, h, kw):
    vcfg, ocfg = _cfgs(vk, oracle, **kw)
    img = vk.gen_synthetic_image(7, w, h)
    with vk.Instance(vcfg) as inst:
        inst.detectFeatures(img, 0)
        pyr = oracle.Pyramid(ocfg, img)
        assert inst.getScaleSpaceNbOctaves() == pyr.nb_octaves
        S = vcfg.nb_scales_per_octave
        for o in range(pyr.nb_octaves):
            assert inst.getScaleSpaceOctaveResolution(o) == pyr.resolution(o)
            for s in range(S + 3):
                g = inst.downloadScaleSpaceImage(o, s)
                ref = pyr.gauss(o, s)
                assert np.array_equal(g.view(np.uint32), ref.view(np.uint32)), (o, s, np.abs(g - ref).max())
            for s in range(S + 2):
                d = inst.downloadDoGImage(o, s)
                ref = pyr.dog(o, s)
                assert np.array_equal(d.view(np.uint32), ref.view(np.uint32)), (o, s, np.abs(d - ref).max())


@pytest.mark.parametrize("w,h,kw", [
    (320, 240, {}),
    (200, 150, {"use_input_upsampling": False}),
    (320, 240, {"descriptor_format": 1, "max_nb_orientation_per_keypoint": 0}),
])
def test_features_bit_exact(vk, oracle, w, h, kw):
    vcfg, ocfg = _cfgs(vk, oracle, **kw)
    img = vk.gen_synthetic_image(11, w, h)
    with vk.Instance(vcfg) as inst:
        inst.detectFeatures(img, 0)
        feats = inst.downloadFeatures(0)
    ref, counts = oracle.detect(ocfg, img)
    assert len(feats) == len(ref) and len(ref) > 50
    for name in ("scale_idx", "octave_idx"):
        assert np.array_equal(feats[name], ref[name]), name
    for name in ("x", "y", "scale_x", "scale_y", "sigma", "orientation", "intensity"):
        assert np.array_equal(feats[name].view(np.uint32), ref[name].view(np.uint32)), name
    assert np.array_equal(feats["descriptor"], ref["descriptor"])


def test_match_bit_exact(vk, oracle):
    a = vk.gen_synthetic_descriptors(1, 1000)
    b = vk.gen_synthetic_descriptors(2, 777)
    b[5] = b[3]            # duplicate rows: equal distances, earlier index must win
    b[1] = b[0]            # tie between b[0] and b[1]: index 1 becomes best (quirk Q7)
    a[10] = b[0]
    fa = np.zeros(len(a), vk.FEATURE_DTYPE)
    fb = np.zeros(len(b), vk.FEATURE_DTYPE)
    fa["descriptor"] = a
    fb["descriptor"] = b
    cfg = vk.default_config()
    with vk.Instance(cfg) as inst:
        inst.uploadFeatures(fa, 0)
        inst.uploadFeatures(fb, 1)
        inst.matchFeatures(0, 1)
        assert inst.getMatchesNumber() == len(a)
        m = inst.downloadMatches()
    ref = oracle.match_2nn(a, b)
    for name in ("idx_a", "idx_b1", "idx_b2"):
        assert np.array_equal(m[name], ref[name]), name
    assert np.array_equal(m["dist_a_b1"].view(np.uint32), ref["dist_a_b1"].view(np.uint32))
    assert np.array_equal(m["dist_a_b2"].view(np.uint32), ref["dist_a_b2"].view(np.uint32))
    assert m["idx_b1"][10] == 1 and m["idx_b2"][10] == 0


def _match_via_api(vk, a, b, max_nb=None):
    fa = np.zeros(len(a), vk.FEATURE_DTYPE)
    fb = np.zeros(len(b), vk.FEATURE_DTYPE)
    fa["descriptor"] = a
    fb["descriptor"] = b
    cfg = vk.default_config(max_nb_sift_per_buffer=max(len(a), len(b), 1000))
    with vk.Instance(cfg) as inst:
        inst.uploadFeatures(fa, 0)
        inst.uploadFeatures(fb, 1)
        inst.matchFeatures(0, 1)
        return inst.downloadMatches()


def _assert_matches_equal(m, ref):
    for name in ("idx_a", "idx_b1", "idx_b2"):
        assert np.array_equal(m[name], ref[name]), (name, np.flatnonzero(m[name] != ref[name])[:10])
    for name in ("dist_a_b1", "dist_a_b2"):
        assert np.array_equal(m[name].view(np.uint32), ref[name].view(np.uint32)), name


@pytest.mark.parametrize("na,nb,seed", [(64, 2, 5), (1, 3, 6), (300, 17, 7), (2049, 1000, 8), (17000, 1501, 9)])
def test_match_shapes(vk, oracle, na, nb, seed):
    """ragged sizes (not multiples of the 16/64 tiles), both kernel instantiations (na > 16384 -> 64 rows/wave)"""
    a = vk.gen_synthetic_descriptors(seed, na)
    b = vk.gen_synthetic_descriptors(seed + 100, nb)
    _assert_matches_equal(_match_via_api(vk, a, b), oracle.match_2nn(a, b))


def test_match_heavy_ties(vk, oracle):
    """few distinct byte values -> many exactly equal distances: the earlier index must win everywhere"""
    rng = np.random.default_rng(3)
    a = (rng.integers(0, 2, (700, 128)) * 255).astype(np.uint8)
    b = (rng.integers(0, 2, (900, 128)) * 255).astype(np.uint8)
    b[::7] = b[0]          # many duplicates, including b[0] == b[7] == ...
    b[1] = b[0]            # quirk Q7 for every row of A
    a[::5] = b[0]
    _assert_matches_equal(_match_via_api(vk, a, b), oracle.match_2nn(a, b))


def test_match_float_collisions(vk, oracle):
    """full-range random bytes: d2 up to 2^23 where different integers share one float sqrt (quirk Q8)"""
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, (512, 128), dtype=np.uint8)
    b = rng.integers(0, 256, (4096, 128), dtype=np.uint8)
    a[:64] = np.where(rng.random((64, 128)) < 0.5, 0, 255).astype(np.uint8)
    b[:512] = np.where(rng.random((512, 128)) < 0.5, 0, 255).astype(np.uint8)
    _assert_matches_equal(_match_via_api(vk, a, b), oracle.match_2nn(a, b))


@pytest.mark.parametrize("na,nb,via_ptr", [(1200, 900, False), (1200, 900, True), (1500, 5000, True), (300, 33000, True), (9000, 700, False), (20000, 2100, True),
                                          (33000, 5000, True), (40000, 300, False), (70001, 130, True), (2600, 2500, False)])
def test_match_ties_in_every_kernel_regime(vk, oracle, na, nb, via_ptr):
    """quirk Q7 (d(b0) == d(b1): index 1 first), duplicate B rows (earlier index first) and exact zero distances in the one-launch
    small kernel and in the stream-decomposed kernel — few row blocks with the piece count at its cap (300 x 33000), runs that
    end one row block and start the next, more workgroups than tiles (70001 x 130) — through the instance (device-side counts)
    and through the device-pointer entry"""
    rng = np.random.default_rng(na + nb)
    a = vk.gen_synthetic_descriptors(na, na)
    b = vk.gen_synthetic_descriptors(nb, nb)
    b[1] = b[0]                                   # Q7 for every row of A
    dup = rng.permutation(np.arange(2, nb))[: nb // 4]
    b[dup] = b[rng.integers(2, nb, len(dup))]     # many duplicate rows anywhere in B (also across chunk borders)
    hit = rng.permutation(na)[: na // 10]
    a[hit] = b[rng.integers(0, nb, len(hit))]     # zero distances, some of them to b0 / b1 and to duplicated rows
    a[::97] = b[0]
    ref = oracle.match_2nn(a, b)
    if via_ptr:
        import torch
        from vulkansift_amd import multigpu
        rec = multigpu.hip_match_fn(torch.from_numpy(a).cuda(), 0, torch.from_numpy(b).cuda())
        torch.cuda.synchronize()
        got = multigpu.records_to_struct(rec.cpu().numpy())
    else:
        got = _match_via_api(vk, a, b)
    _assert_matches_equal(got, ref)
    assert (ref["idx_b1"][::97] == 1).all() and (ref["idx_b2"][::97] == 0).all()


def test_match_fewer_than_two_b_rows(vk, oracle):
    """nb < 2: this build defines the rows the shader would read as stale memory as zero descriptors"""
    a = vk.gen_synthetic_descriptors(21, 40)
    b = vk.gen_synthetic_descriptors(22, 1)
    m = _match_via_api(vk, a, b)
    ref = oracle.match_2nn(a, np.vstack([b, np.zeros((1, 128), np.uint8)]))
    _assert_matches_equal(m, ref)


def test_match_largest_regime(vk, oracle):
    """many row blocks against a two-tile B: runs of one or two tiles"""
    a = vk.gen_synthetic_descriptors(41, 33001)
    b = vk.gen_synthetic_descriptors(42, 130)
    _assert_matches_equal(_match_via_api(vk, a, b), oracle.match_2nn(a, b))


def test_match_device_pointer_api_chunked(vk, oracle):
    """vksift_hip_match_2nn_desc on torch tensors: the stream-decomposed path with ties across piece borders"""
    import torch
    from vulkansift_amd import multigpu
    a = vk.gen_synthetic_descriptors(51, 33000)
    b = vk.gen_synthetic_descriptors(52, 1000)
    b[1] = b[0]
    b[700] = b[3]          # duplicate in a later chunk: the earlier index must win
    b[999] = b[130]
    a[7] = b[3]
    a[8] = b[130]
    rec = multigpu.hip_match_fn(torch.from_numpy(a).cuda(), 100, torch.from_numpy(b).cuda())
    torch.cuda.synchronize()
    got = multigpu.records_to_struct(rec.cpu().numpy())
    ref = oracle.match_2nn(a, b)
    assert np.array_equal(got["idx_a"], ref["idx_a"] + 100)
    for name in ("idx_b1", "idx_b2"):
        assert np.array_equal(got[name], ref[name]), name
    assert np.array_equal(got["dist_a_b1"].view(np.uint32), ref["dist_a_b1"].view(np.uint32))
    assert got["idx_b1"][7] == 3 and got["idx_b2"][7] == 700


def test_match_device_pointer_api_refuses_an_undersized_scratch(vk):
    """the scratch requirement of vksift_hip_match_2nn_desc grew with the cell scan (ABI version 5): a buffer sized by the old
    formula (2 na + nb + 5 * 32 na words) is refused with hipErrorInvalidValue and NOTHING is launched — the record buffer keeps
    its fill pattern; the library's own figure is accepted"""
    import torch
    L = vk.lib()
    assert L.vksift_hip_abi_version() >= 6
    na = nb = 40000
    a = torch.from_numpy(vk.gen_synthetic_descriptors(71, na)).cuda()
    b = torch.from_numpy(vk.gen_synthetic_descriptors(72, nb)).cuda()
    out = torch.full((na, 5), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    need = int(L.vksift_hip_match_scratch_u32(na, nb))
    old = 2 * na + nb + 5 * na * 32
    assert old < need
    scratch = torch.empty(need, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    err = L.vksift_hip_match_2nn_desc(a.data_ptr(), na, 0, b.data_ptr(), nb, scratch.data_ptr(), old, out.data_ptr(), stream)
    torch.cuda.synchronize()
    assert err == 1 and bool((out == 0x5A5A5A5A).all())
    err = L.vksift_hip_match_2nn_desc(a.data_ptr(), na, 0, b.data_ptr(), nb, scratch.data_ptr(), need, out.data_ptr(), stream)
    torch.cuda.synchronize()
    assert err == 0 and bool((out[:, 0].cpu() == torch.arange(na, dtype=torch.int32)).all())


@pytest.mark.parametrize("na,nb,via_ptr", [(9000, 300, False), (33000, 520, True)])
def test_match_float_collisions_all_regimes(vk, oracle, na, nb, via_ptr):
    """d2 >= 2^22 candidates in the 16-rows-per-wave and the B-chunked kernels: flagged rows are replayed by the exact
    scalar kernel (quirk Q8), all other rows keep the integer MFMA result"""
    rng = np.random.default_rng(na)
    a = vk.gen_synthetic_descriptors(61, na)
    b = vk.gen_synthetic_descriptors(62, nb)
    ext = slice(0, na, 37)
    a[ext] = np.where(rng.random((len(range(na)[ext]), 128)) < 0.5, 0, 255).astype(np.uint8)
    b[::3] = np.where(rng.random((len(range(nb)[::3]), 128)) < 0.5, 0, 255).astype(np.uint8)
    b[1] = 255 - b[0] // 255 * 255   # first two columns far apart / extreme for some rows
    ref = oracle.match_2nn(a, b)
    if via_ptr:
        import torch
        from vulkansift_amd import multigpu
        rec = multigpu.hip_match_fn(torch.from_numpy(a).cuda(), 0, torch.from_numpy(b).cuda())
        torch.cuda.synchronize()
        got = multigpu.records_to_struct(rec.cpu().numpy())
    else:
        got = _match_via_api(vk, a, b)
    _assert_matches_equal(got, ref)


def test_full_size_1080p_detection_bit_exact(vk, oracle):
    """BASELINE config 3 resolution (1920x1080, up-sampling ON: 7 octaves, 3840x2160 octave 0): every feature bit-exact"""
    w, h = 1920, 1080
    img = vk.gen_synthetic_image(77, w, h)
    vcfg, ocfg = _cfgs(vk, oracle, input_image_max_size=w * h)
    with vk.Instance(vcfg) as inst:
        inst.detectFeatures(img, 0)
        feats = inst.downloadFeatures(0)
        assert inst.getScaleSpaceNbOctaves() == 7
    ref, _ = oracle.detect(ocfg, img)
    assert len(feats) == len(ref) and len(ref) > 5000
    assert feats.tobytes() == ref.tobytes()


def test_full_size_50k_matcher_properties_and_samples(vk, oracle):
    """BASELINE config 4 size (50k x 50k, B-chunked MFMA kernel): self-match property on every row + oracle on sampled rows"""
    import torch
    from vulkansift_amd import multigpu
    n = 50000
    a = vk.gen_synthetic_descriptors(91, n)
    b = vk.gen_synthetic_descriptors(92, n)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    # property: A against itself -> every row's best match is itself (or an earlier identical row) at distance 0
    got = multigpu.records_to_struct(multigpu.hip_match_fn(da, 0, da).cpu().numpy())
    assert np.all(got["dist_a_b1"] == 0.0)
    assert np.all(got["idx_b1"] <= np.arange(n))
    same = np.all(a[got["idx_b1"]] == a, axis=1)
    assert same.all()
    assert np.all(got["dist_a_b2"] >= got["dist_a_b1"])
    # oracle on a sample of rows against the full B
    got = multigpu.records_to_struct(multigpu.hip_match_fn(da, 0, db).cpu().numpy())
    rows = np.random.default_rng(1).choice(n, 300, replace=False)
    ref = oracle.match_2nn(a[rows], b)
    for name in ("idx_b1", "idx_b2"):
        assert np.array_equal(got[name][rows], ref[name]), name
    assert np.array_equal(got["dist_a_b1"][rows].view(np.uint32), ref["dist_a_b1"].view(np.uint32))
    assert np.array_equal(got["dist_a_b2"][rows].view(np.uint32), ref["dist_a_b2"].view(np.uint32))


def test_randomised_detection_parity(vk, oracle):
    """random resolutions x random configurations (seeded): every feature byte equals the oracle's"""
    rng = np.random.default_rng(2024)
    for case in range(36):
        w, h = int(rng.integers(48, 420)), int(rng.integers(48, 320))
        kw = {}
        if rng.random() < 0.4:
            kw["use_input_upsampling"] = False
        if rng.random() < 0.3:
            kw["use_hardware_interpolated_blur"] = False
        if rng.random() < 0.4:
            kw["nb_scales_per_octave"] = int(rng.choice([2, 4, 5]))
        if rng.random() < 0.3:
            kw["descriptor_format"] = 1
        if rng.random() < 0.3:
            kw["max_nb_orientation_per_keypoint"] = int(rng.choice([0, 1, 2]))
        if rng.random() < 0.3:
            kw["nb_octaves"] = int(rng.integers(1, 4))
        if rng.random() < 0.3:
            kw["intensity_threshold"] = float(rng.choice([0.02, 0.06]))
        vcfg, ocfg = _cfgs(vk, oracle, **kw)
        img = vk.gen_synthetic_image(1000 + case, w, h)
        with vk.Instance(vcfg) as inst:
            inst.detectFeatures(img, 0)
            feats = inst.downloadFeatures(0)
        ref, _ = oracle.detect(ocfg, img)
        assert len(feats) == len(ref), (case, w, h, kw, len(feats), len(ref))
        assert feats.tobytes() == ref.tobytes(), (case, w, h, kw)


def test_randomised_matcher_parity(vk, oracle):
    """random (N_A, N_B) incl. tiny and ragged sizes, duplicates and near-duplicates: all five record fields bit-exact"""
    rng = np.random.default_rng(77)
    for case in range(30):
        na, nb = int(rng.integers(1, 2600)), int(rng.integers(2, 2600))
        a = vk.gen_synthetic_descriptors(2000 + case, na)
        b = vk.gen_synthetic_descriptors(3000 + case, nb)
        k = int(rng.integers(0, min(na, nb) // 2 + 1))
        if k:
            idx = rng.permutation(nb)[:k]
            b[idx] = np.clip(a[:k].astype(np.int32) + rng.integers(-3, 4, (k, 128)), 0, 255).astype(np.uint8)
            b[idx[: k // 3]] = a[: k // 3]           # exact duplicates -> zero distances and ties
        if nb > 4 and rng.random() < 0.5:
            b[1] = b[0]                               # quirk Q7
        _assert_matches_equal(_match_via_api(vk, a, b), oracle.match_2nn(a, b))


def test_randomised_detection_parity_wide(vk, oracle):
    """second distribution: larger images, continuous parameters (sigmas, thresholds), batches of random size"""
    rng = np.random.default_rng(4242)
    for case in range(12):
        w, h = int(rng.integers(200, 900)), int(rng.integers(150, 640))
        kw = {"seed_scale_sigma": float(np.float32(rng.uniform(1.3, 2.6))), "input_image_blur_level": float(np.float32(rng.uniform(0.3, 0.6))),
              "intensity_threshold": float(np.float32(rng.uniform(0.015, 0.08))), "edge_threshold": float(np.float32(rng.uniform(5.0, 15.0)))}
        if rng.random() < 0.5:
            kw["use_input_upsampling"] = False
        if rng.random() < 0.5:
            kw["nb_scales_per_octave"] = int(rng.integers(1, 7))
        if rng.random() < 0.3:
            kw["use_hardware_interpolated_blur"] = False
        nb = int(rng.integers(1, 4))
        vcfg, ocfg = _cfgs(vk, oracle, input_image_max_size=w * h, **kw)
        vcfg.sift_buffer_count = nb
        imgs = [vk.gen_synthetic_image(5000 + 10 * case + i, w, h) for i in range(nb)]
        with vk.Instance(vcfg, batch_capacity=nb) as inst:
            inst.detectFeaturesBatch(imgs, 0)
            feats = [inst.downloadFeatures(i) for i in range(nb)]
        for i in range(nb):
            ref, _ = oracle.detect(ocfg, imgs[i])
            assert len(feats[i]) == len(ref), (case, i, w, h, kw, len(feats[i]), len(ref))
            assert feats[i].tobytes() == ref.tobytes(), (case, i, w, h, kw)


def test_reference_large_image_size_bit_exact(vk, oracle):
    """3456x2304, the largest size of the reference's published benchmark (docs/Performances.md:27): 8 octaves, ~50 k features"""
    w, h = 3456, 2304
    img = vk.gen_synthetic_image(3456, w, h)
    vcfg, ocfg = _cfgs(vk, oracle, input_image_max_size=w * h)
    with vk.Instance(vcfg) as inst:
        inst.detectFeatures(img, 0)
        feats = inst.downloadFeatures(0)
    ref, _ = oracle.detect(ocfg, img)
    assert len(feats) == len(ref) and len(ref) > 30000
    assert feats.tobytes() == ref.tobytes()


@pytest.mark.parametrize("nb,w,h,kw", [(9, 256, 192, {}), (16, 208, 160, {"use_input_upsampling": False, "nb_scales_per_octave": 2}),
                                       (11, 320, 200, {"max_nb_sift_per_buffer": 300})])
def test_large_batch_detection_and_self_match_bit_exact(vk, oracle, nb, w, h, kw):
    """batches of 8 images and more take their own launch shapes (estimated feature-stage grids, image-fastest work order,
    slot-fastest batched matcher): every image against the oracle, including a buffer capacity small enough to clamp"""
    vcfg, ocfg = _cfgs(vk, oracle, input_image_max_size=w * h, **kw)
    vcfg.sift_buffer_count = nb
    imgs = [vk.gen_synthetic_image(7000 + i, w, h) for i in range(nb)]
    with vk.Instance(vcfg, batch_capacity=nb) as inst:
        inst.detectFeaturesBatch(imgs, 0)
        feats = [inst.downloadFeatures(i) for i in range(nb)]
        pairs = [(i, (i + 1) % nb) for i in range(nb)]
        inst.matchFeaturesBatch([a for a, _ in pairs], [b for _, b in pairs])
        got = [inst.downloadMatchesBatch(k) for k in range(nb)]
    refs = [oracle.detect(ocfg, im)[0] for im in imgs]
    for i in range(nb):
        assert len(feats[i]) == len(refs[i]), (i, len(feats[i]), len(refs[i]))
        assert feats[i].tobytes() == refs[i].tobytes(), i
    for (a, b), m in zip(pairs, got):
        _assert_matches_equal(m, oracle.match_2nn(refs[a], refs[b]))


@pytest.mark.parametrize("w,h", [(139, 356), (64, 800), (800, 64), (1100, 90)])
def test_narrow_images_within_the_configured_area_are_accepted(vk, oracle, w, h):
    """any w x h <= input_image_max_size must work (the reference re-creates its images per resolution, sift_memory.c:362-452):
    the row padding of a narrow image needs more scratch than the square reservation, which then grows on demand"""
    vcfg, ocfg = _cfgs(vk, oracle, input_image_max_size=w * h)
    img = vk.gen_synthetic_image(w * 7 + h, w, h)
    sq = vk.gen_synthetic_image(5, 128, 128)
    with vk.Instance(vcfg) as inst:
        inst.detectFeatures(sq, 0)                 # a square image first: the reservation is in use, then outgrown
        inst.detectFeatures(img, 0)
        feats = inst.downloadFeatures(0)
        top = inst.downloadScaleSpaceImage(0, 1)
        inst.detectFeatures(sq, 0)                 # and back
        again = inst.downloadFeatures(0)
    ref, _ = oracle.detect(ocfg, img)
    assert feats.tobytes() == ref.tobytes()
    assert np.isfinite(top).all()
    assert again.tobytes() == oracle.detect(ocfg, sq)[0].tobytes()


@pytest.mark.parametrize("w,h,ups", [(41, 29, False), (16, 64, True), (16, 70, False), (33, 70, False), (31, 64, False), (24, 48, True)])
def test_images_too_small_for_an_octave_detect_nothing(vk, oracle, w, h, ups):
    """nb_octaves = log2(shortest side) - 4 (+1 with up-sampling) (sift_memory.c:22): zero octaves is an empty result, not an
    error; one octave of a tiny image must match the oracle like any other"""
    vcfg, ocfg = _cfgs(vk, oracle, input_image_max_size=128 * 128, use_input_upsampling=ups)
    img = vk.gen_synthetic_image(w + 3 * h, w, h)
    with vk.Instance(vcfg) as inst:
        inst.detectFeatures(img, 0)
        n = inst.getFeaturesNumber(0)
        feats = inst.downloadFeatures(0)
        inst.detectFeatures(img, 1)
        inst.matchFeatures(0, 1)
        m = inst.downloadMatches()
    ref, _ = oracle.detect(ocfg, img)
    assert n == len(ref) and feats.tobytes() == ref.tobytes()
    assert len(m) == len(ref)
